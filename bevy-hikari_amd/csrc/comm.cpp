// comm.cpp - the halo exchanges of a band-sharded frame, inside the boundary.
//
// The reference renders a frame from one process into one command encoder (src/light.rs:590-702,
// src/post_process.rs:1131-1311).  Sharded over the GPUs of a node the same dispatch order needs
// neighbour exchanges between the temporal and spatial dispatches (light.rs:689-697) and before the denoiser; which
// rows of which buffer is hk_band_plan_for / hk_band_schedule (host_logic.cpp).  This file executes that schedule:
//
//   hk_comm_*   one process per GPU: ncclSend / ncclRecv pairs inside one group per exchange, on the context's stream
//               (RCCL over xGMI; point-to-point, neighbouring bands only).  librccl is dlopen'ed on first use.
//   hk_multi_*  one process, several GPUs (what a Bevy render thread would drive): hipMemcpyPeerAsync on the receiver's
//               stream, ordered against the owner's stream with events.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is resolved at run time
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hk_internal.hpp"

using namespace hk;

#define HK_HIP(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      ::hk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);        \
      return HK_E_HIP;                                                                                   \
    }                                                                                                    \
  } while (0)

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommSplit) CommSplit = nullptr;   // optional (RCCL >= 2.18): the second communicator of a context
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;  // contexts may be initialised from different threads
  std::call_once(once, [] {
    // a copy some other component of the process already mapped (PyTorch ships its own) is as good as the system one
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (r.lib) {
#define HK_SYM(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.lib, "nccl" #name))
      HK_SYM(GetUniqueId); HK_SYM(CommInitRank); HK_SYM(CommDestroy); HK_SYM(GroupStart); HK_SYM(GroupEnd); HK_SYM(Send); HK_SYM(Recv); HK_SYM(GetErrorString);
      HK_SYM(CommSplit);
#undef HK_SYM
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
        dlclose(r.lib);
        r.lib = nullptr;
      }
    }
  });
  return r.lib ? &r : nullptr;
}

#define HK_NCCL(R, expr)                                                                     \
  do {                                                                                       \
    ncclResult_t e_ = (expr);                                                                \
    if (e_ != ncclSuccess) {                                                                 \
      ::hk::set_error("%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(e_), __FILE__, __LINE__); \
      return HK_E_HIP;                                                                       \
    }                                                                                        \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  uint32_t rank = 0, n_ranks = 1;
  // schedules are cached per (stage argument, frame parity, settings): a band's frame is a few hundred microseconds at
  // 8 GPUs, so nothing is re-planned per frame
  struct Cached {
    uint32_t stage_arg, rank, parity, width, height, n_ranks, bounds_generation;
    float ratio;
    HkSettings st;
    std::vector<HkTransfer> tr;
  };
  std::deque<Cached> cache;  // (deque: references stay valid while entries are appended)
  uint64_t exchanges = 0, bytes_received = 0;
  // Every RCCL call of this context goes to ONE stream of its own, ordered against the context's stream with events: an exchange
  // waits for what the context has enqueued (`ready`) and the context's stream waits for the exchange (`done`) - the order a
  // single stream would give - while the GATHER of a finished frame makes nobody wait: the next frame renders while rank 0
  // collects the rows (`gather_done`; comm_join before the plane is written again, or before anybody reads the image).
  // Round 6: TWO lanes.  Lane 0 carries what a band's main stream waits for - exchanges C and A, a migration; lane 1 what its
  // post-processing and the overlay wait for - exchange B, the gather, exchanges D / E.  Operations on one communicator execute in
  // issue order: on a single lane frame n's exchange B and gather (which wait for frame n's post-processing) would sit in front of
  // frame n + 1's exchange A, and the main stream would wait for the post stream after all.  Lane 1 is a communicator of its own
  // (ncclCommSplit of the first, the same ranks) on a stream of its own; where the library cannot split, lane 1 IS lane 0 (correct,
  // only serialised as before).  All ranks issue the same sequence of calls per lane.
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr, gather_done = nullptr;
  ncclComm_t comm2 = nullptr;
  hipStream_t stream2 = nullptr;
  hipEvent_t ready2 = nullptr, done2 = nullptr;
  bool gather_pending = false;
  uint32_t gather_parity = 0;  // frame parity of the plane a pending gather reads / fills
};

int schedule_for(std::deque<Comm::Cached>& cache, const CtxInfo& ci, uint32_t rank, uint32_t n_ranks, uint32_t stage_arg, const HkSettings* st,
                 const std::vector<HkTransfer>** out) {
  for (const Comm::Cached& e : cache)
    if (e.stage_arg == stage_arg && e.rank == rank && e.n_ranks == n_ranks && e.bounds_generation == ci.bounds_generation && e.parity == (ci.frame_number & 1u) &&
        e.width == ci.width && e.height == ci.height && e.ratio == ci.ratio && memcmp(&e.st, st, sizeof(HkSettings)) == 0) {
      *out = &e.tr;
      return HK_OK;
    }
  Comm::Cached e;
  e.stage_arg = stage_arg;
  e.rank = rank;
  e.n_ranks = n_ranks;
  e.bounds_generation = ci.bounds_generation;
  e.parity = ci.frame_number & 1u;
  e.width = ci.width;
  e.height = ci.height;
  e.ratio = ci.ratio;
  e.st = *st;
  uint32_t n = 0;
  int rc = hk_band_schedule_bounds(ci.width, ci.height, ci.ratio, ci.band_bounds, rank, n_ranks, stage_arg, ci.frame_number, st, nullptr, &n);
  if (rc) return rc;
  e.tr.resize(n);
  if (n && (rc = hk_band_schedule_bounds(ci.width, ci.height, ci.ratio, ci.band_bounds, rank, n_ranks, stage_arg, ci.frame_number, st, e.tr.data(), &n))) return rc;
  cache.push_back(std::move(e));  // (bounded by the callers: they clear the cache when it grows past a few hundred entries)
  *out = &cache.back().tr;
  return HK_OK;
}

// One exchange: every transfer of the list as an ncclSend / ncclRecv on `stream`, inside ONE group (the sends and receives of a
// rank pair up across ranks by issue order; inside a group nothing blocks before ncclGroupEnd).
int run_transfers(hk_ctx* c, Comm* cm, Rccl* R, const HkTransfer* tr, size_t n, hipStream_t stream, ncclComm_t comm) {
  HK_NCCL(R, R->GroupStart());
  for (size_t k = 0; k < n; ++k) {
    const HkTransfer& t = tr[k];
    size_t logical = 0;
    char* base = static_cast<char*>(ctx_buffer(c, t.buffer, &logical));
    if (!base || t.offset + t.bytes > logical) {
      (void)R->GroupEnd();
      HK_REQUIRE(false, HK_E_INVALID, "halo transfer outside buffer %u", t.buffer);
    }
    ncclResult_t e = t.is_recv ? R->Recv(base + t.offset, t.bytes, ncclUint8, (int)t.peer, comm, stream)
                               : R->Send(base + t.offset, t.bytes, ncclUint8, (int)t.peer, comm, stream);
    if (e != ncclSuccess) {
      (void)R->GroupEnd();
      set_error("ncclSend/ncclRecv failed: %s", R->GetErrorString(e));
      return HK_E_HIP;
    }
    if (t.is_recv) cm->bytes_received += t.bytes;
  }
  HK_NCCL(R, R->GroupEnd());
  return HK_OK;
}

// the transfers on the communicator's stream, behind everything `main` holds; wait = `main` continues only after them
// lane 0: the communicator's first stream (exchanges the band's main stream waits for); lane 1: the second communicator and stream
// (exchange B, the gather, exchanges D / E) where the library could split, else lane 0
int run_ordered(hk_ctx* c, Comm* cm, Rccl* R, const HkTransfer* tr, size_t n, hipStream_t main, bool wait, int lane = 0) {
  const bool second = lane == 1 && cm->comm2 && cm->stream2;
  hipStream_t stream = second ? cm->stream2 : cm->stream;
  hipEvent_t ready = second ? cm->ready2 : cm->ready, done = second ? cm->done2 : cm->done;
  HK_HIP(hipEventRecord(ready, main));
  HK_HIP(hipStreamWaitEvent(stream, ready, 0));
  const int rc = run_transfers(c, cm, R, tr, n, stream, second ? cm->comm2 : cm->comm);
  if (rc) return rc;
  HK_HIP(hipEventRecord(wait ? done : cm->gather_done, stream));
  if (wait) HK_HIP(hipStreamWaitEvent(main, done, 0));
  return HK_OK;
}

}  // namespace

namespace hk {

// the context's stream waits for a gather still in flight: parity < 0 = whatever is pending (somebody is about to look at the image),
// else only a gather of that frame parity's plane (the frame being begun will write it)
int comm_join(hk_ctx* c, int parity) {
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  if (!cm || !cm->gather_pending || (parity >= 0 && (uint32_t)parity != cm->gather_parity)) return HK_OK;
  CtxInfo ci;
  const int rc = ctx_info(c, &ci);
  if (rc) return rc;
  HK_HIP(hipStreamWaitEvent((hipStream_t)ci.stream, cm->gather_done, 0));
  cm->gather_pending = false;
  return HK_OK;
}

int comm_exchange(hk_ctx* c, uint32_t stage_arg, const HkSettings* st) {
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  HK_REQUIRE(cm && cm->comm, HK_E_NOT_READY, "no communicator attached (hk_comm_init)");
  HK_REQUIRE(st, HK_E_INVALID, "settings is NULL");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded");
  CtxInfo ci;
  int rc = ctx_info(c, &ci);
  if (rc) return rc;
  HK_REQUIRE(ci.width > 0, HK_E_NOT_READY, "hk_resize has not been called");
  const std::vector<HkTransfer>* tr = nullptr;
  if (cm->cache.size() > 256) cm->cache.clear();  // settings changed many times: start over
  if ((rc = schedule_for(cm->cache, ci, cm->rank, cm->n_ranks, stage_arg, st, &tr))) return rc;
  if (tr->empty()) return HK_OK;
  HK_HIP(hipSetDevice(ci.device));
  // exchanges D / E read what the a-trous levels + tone mapping left on the post stream (frame pipelining)
  if ((stage_arg & 0xffu) >= HK_STAGE_ANTIALIAS && (rc = ctx_join_side(c))) return rc;
  // No join with the side stream here: hk_frame_stage joins it exactly where an exchange reads what the direct-light
  // dispatches wrote (end of TEMPORAL when the emissive channel has a spatial pass, end of SPATIAL before exchange B), so
  // exchange A (indirect reservoirs, main stream) overlaps the direct-light kernels still running on the side stream.
  if ((rc = run_ordered(c, cm, R, tr->data(), tr->size(), (hipStream_t)ci.stream, true, (stage_arg & 0xffu) >= HK_STAGE_POST_PROCESS ? 1 : 0))) return rc;
  cm->exchanges += 1;
  return HK_OK;
}

// hk_migrate_bands: the rows of the history reservoirs that change owner between two splits, as one exchange in stream order
int comm_migrate(hk_ctx* c, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t next_frame_number, const HkSettings* st) {
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  HK_REQUIRE(cm && cm->comm, HK_E_NOT_READY, "no communicator attached (hk_comm_init): a host with its own transport moves hk_band_migration_schedule's rows itself");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded");
  CtxInfo ci;
  int rc = ctx_info(c, &ci);
  if (rc) return rc;
  uint32_t n = 0;
  if ((rc = hk_band_migration_schedule(ci.width, ci.height, ci.ratio, old_bounds, new_bounds, cm->rank, cm->n_ranks, next_frame_number, st, nullptr, &n))) return rc;
  std::vector<HkTransfer> tr(n);
  if (n && (rc = hk_band_migration_schedule(ci.width, ci.height, ci.ratio, old_bounds, new_bounds, cm->rank, cm->n_ranks, next_frame_number, st, tr.data(), &n))) return rc;
  if (tr.empty()) return HK_OK;
  HK_HIP(hipSetDevice(ci.device));
  if ((rc = run_ordered(c, cm, R, tr.data(), tr.size(), (hipStream_t)ci.stream, true))) return rc;
  cm->exchanges += 1;
  return HK_OK;
}

// the root collects every other band's rows of `buffer` (ncclSend / ncclRecv pairs in one group, on the stream)
int comm_gather(hk_ctx* c, uint32_t buffer, uint32_t root, bool overlap) {
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  HK_REQUIRE(cm && cm->comm, HK_E_NOT_READY, "no communicator attached (hk_comm_init)");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded");
  CtxInfo ci;
  int rc = ctx_info(c, &ci);
  if (rc) return rc;
  HK_REQUIRE(ci.width > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(root < cm->n_ranks, HK_E_INVALID, "root %u of %u ranks", root, cm->n_ranks);
  if (cm->n_ranks < 2) return HK_OK;
  uint32_t n = 0;
  if ((rc = hk_band_gather_schedule(ci.width, ci.height, ci.ratio, ci.upscale_kind, ci.band_bounds, cm->rank, cm->n_ranks, root, buffer, nullptr, &n))) return rc;
  std::vector<HkTransfer> tr(n);
  if (n && (rc = hk_band_gather_schedule(ci.width, ci.height, ci.ratio, ci.upscale_kind, ci.band_bounds, cm->rank, cm->n_ranks, root, buffer, tr.data(), &n))) return rc;
  if (tr.empty()) return HK_OK;
  HK_HIP(hipSetDevice(ci.device));
  // overlap: nobody waits - the next frame writes the OTHER parity's plane, and the frame after it (or whoever reads the image
  // first) joins.  Only for a plane that is double-buffered by frame parity and that the next frame does not read.
  overlap = overlap && buffer == HK_BUF_TONE_MAPPED;
  // (one pending gather is tracked: an older one still in flight - of either parity - is joined before its record is overwritten;
  // hk_frame_render joins everything before it gathers, a caller of hk_comm_gather alone need not have)
  if (overlap && (rc = comm_join(c, -1))) return rc;
  if ((rc = run_ordered(c, cm, R, tr.data(), tr.size(), (hipStream_t)ci.stream, !overlap, 1))) return rc;
  if (overlap) {
    cm->gather_pending = true;
    cm->gather_parity = ci.frame_number & 1u;
  }
  return HK_OK;
}

void comm_release(hk_ctx* c) {
  void** slot = ctx_comm_slot(c);
  Comm* cm = static_cast<Comm*>(*slot);
  if (!cm) return;
  Rccl* R = rccl();
  if (cm->stream) (void)hipStreamSynchronize(cm->stream);
  if (cm->stream2) (void)hipStreamSynchronize(cm->stream2);
  if (R && cm->comm2) (void)R->CommDestroy(cm->comm2);
  if (R && cm->comm) (void)R->CommDestroy(cm->comm);
  for (hipEvent_t e : {cm->ready, cm->done, cm->gather_done, cm->ready2, cm->done2})
    if (e) (void)hipEventDestroy(e);
  if (cm->stream) (void)hipStreamDestroy(cm->stream);
  if (cm->stream2) (void)hipStreamDestroy(cm->stream2);
  delete cm;
  *slot = nullptr;
}

}  // namespace hk

// ------------------------------------------------------------------ hk_multi
struct MultiPool;
static std::atomic<int> g_multi_serial{0};  // hk_debug_multi_serial: the calling thread enqueues every band (the A/B of the enqueue threads)
struct hk_multi {
  std::vector<hk_ctx*> ctx;
  std::vector<int> device;
  std::vector<hipEvent_t> produced, copied;  // per context: "my stage is enqueued up to here" / "my incoming copies are enqueued up to here"
  std::deque<Comm::Cached> cache;
  uint64_t bytes_copied = 0;
  MultiPool* pool = nullptr;  // one enqueue thread per band (hk_multi_frame_render), created on first use
};

// A frame of n bands is n x (10-25 launches + event traffic) of host work: enqueued band after band by the calling thread it
// costs n x ~60 us - more than a 135-row band's GPU time at 8 GPUs.  Each band therefore has a thread
// of its own that enqueues ITS stages and ITS side of the exchanges; the threads meet at two spin barriers per exchange
// (every "produced" event is recorded before anybody waits for one; every "copied" event before its owner waits for it), so
// an event is never re-recorded while some other thread still means its previous recording.  The GPU-side order is the
// serial path's (multi_exchange below): same events, same waits, same copies.  HK_MULTI_SERIAL=1 keeps the one-thread path.
struct MultiJob {
  const HkFrame* f = nullptr;
  const HkView* v = nullptr;
  const HkPreviousView* pv = nullptr;
  const HkLights* l = nullptr;
  const HkSettings* st = nullptr;
  uint32_t flags = 0;
};
struct MultiPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  uint64_t job_id = 0;
  uint32_t done = 0;
  bool quit = false;
  MultiJob job;
  std::vector<int> rc;
  std::vector<std::string> err;
  std::vector<std::deque<Comm::Cached>> cache;  // per band: schedule_for is not thread-safe on a shared cache
  std::vector<uint64_t> bytes;
  // spin barrier (sense = generation); `failed` releases everybody
  std::atomic<uint32_t> arrived{0}, generation{0};
  std::atomic<int> failed{0};
  uint32_t n = 0;
  bool barrier() {
    const uint32_t gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      arrived.store(0, std::memory_order_relaxed);
      generation.store(gen + 1, std::memory_order_release);
      return !failed.load(std::memory_order_acquire);
    }
    for (uint32_t spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins) {
      if (failed.load(std::memory_order_acquire)) return false;
      if (spins > 2000) std::this_thread::yield();
    }
    return !failed.load(std::memory_order_acquire);
  }
};

namespace {

// the receive side of one transfer list per band as peer copies on the receiver's stream, ordered by events against the owners' streams
int multi_run_plans(hk_multi* m, const std::vector<const std::vector<HkTransfer>*>& plans, const std::vector<CtxInfo>& ci) {
  const uint32_t n = (uint32_t)m->ctx.size();
  // 1. every band: "produced" recorded on its stream (hk_frame_stage has joined the side stream wherever an exchange reads
  //    what the direct-light dispatches wrote; exchange A therefore overlaps them)
  for (uint32_t i = 0; i < n; ++i) {
    HK_HIP(hipSetDevice(m->device[i]));
    HK_HIP(hipEventRecord(m->produced[i], (hipStream_t)ci[i].stream));
  }
  // 2. every band: wait for the owners of the rows it receives, then copy them in
  std::vector<uint8_t> read_by(n * n, 0);  // read_by[p * n + i]: band i copies rows out of band p
  for (uint32_t i = 0; i < n; ++i) {
    HK_HIP(hipSetDevice(m->device[i]));
    hipStream_t stream = (hipStream_t)ci[i].stream;
    std::vector<uint8_t> waited(n, 0);
    bool copies = false;
    for (const HkTransfer& t : *plans[i]) {
      if (!t.is_recv) continue;
      HK_REQUIRE(t.peer < n, HK_E_INVALID, "bad peer in schedule");
      if (!waited[t.peer]) {
        HK_HIP(hipStreamWaitEvent(stream, m->produced[t.peer], 0));
        waited[t.peer] = 1;
      }
      size_t ldst = 0, lsrc = 0;
      char* dst = static_cast<char*>(ctx_buffer(m->ctx[i], t.buffer, &ldst));
      const char* src = static_cast<const char*>(ctx_buffer(m->ctx[t.peer], t.buffer, &lsrc));
      HK_REQUIRE(dst && src && t.offset + t.bytes <= ldst && t.offset + t.bytes <= lsrc, HK_E_INVALID, "halo transfer outside buffer %u", t.buffer);
      if (m->device[i] == m->device[t.peer])
        HK_HIP(hipMemcpyAsync(dst + t.offset, src + t.offset, t.bytes, hipMemcpyDeviceToDevice, stream));
      else
        HK_HIP(hipMemcpyPeerAsync(dst + t.offset, m->device[i], src + t.offset, m->device[t.peer], t.bytes, stream));
      read_by[t.peer * n + i] = 1;
      m->bytes_copied += t.bytes;
      copies = true;
    }
    if (copies) HK_HIP(hipEventRecord(m->copied[i], stream));
  }
  // 3. an owner does not run ahead of the copies that read its rows (its next dispatches may overwrite them)
  for (uint32_t p = 0; p < n; ++p) {
    HK_HIP(hipSetDevice(m->device[p]));
    for (uint32_t i = 0; i < n; ++i)
      if (read_by[p * n + i]) HK_HIP(hipStreamWaitEvent((hipStream_t)ci[p].stream, m->copied[i], 0));
  }
  return HK_OK;
}

// everything band i receives before `stage_arg`, as peer copies on i's stream; events order them against the owners' streams
int multi_exchange(hk_multi* m, uint32_t stage_arg, const HkSettings* st) {
  const uint32_t n = (uint32_t)m->ctx.size();
  if (n < 2) return HK_OK;
  std::vector<const std::vector<HkTransfer>*> plans(n, nullptr);
  std::vector<CtxInfo> ci(n);
  bool any = false;
  if (m->cache.size() > 1024) m->cache.clear();
  for (uint32_t i = 0; i < n; ++i) {
    int rc = ctx_info(m->ctx[i], &ci[i]);
    if (rc) return rc;
    HK_REQUIRE(ci[i].width > 0, HK_E_NOT_READY, "hk_multi_resize has not been called");
    if ((rc = schedule_for(m->cache, ci[i], i, n, stage_arg, st, &plans[i]))) return rc;
    any = any || !plans[i]->empty();
  }
  if (!any) return HK_OK;
  if ((stage_arg & 0xffu) >= HK_STAGE_ANTIALIAS)  // exchanges D / E read what the post stream wrote (frame pipelining)
    for (uint32_t i = 0; i < n; ++i) {
      const int rj = ctx_join_side(m->ctx[i]);
      if (rj) return rj;
    }
  return multi_run_plans(m, plans, ci);
}

// band i's side of multi_exchange, on band i's thread.  Returns HK_OK or the error; false from a barrier = some other band failed.
int band_exchange(hk_multi* m, uint32_t i, uint32_t stage_arg, const HkSettings* st) {
  MultiPool* P = m->pool;
  const uint32_t n = (uint32_t)m->ctx.size();
  CtxInfo ci;
  int rc = ctx_info(m->ctx[i], &ci);
  const std::vector<HkTransfer>* plan = nullptr;
  if (!rc) {
    if (P->cache[i].size() > 256) P->cache[i].clear();
    rc = schedule_for(P->cache[i], ci, i, n, stage_arg, st, &plan);
  }
  hipStream_t stream = (hipStream_t)ci.stream;
  if (!rc && (stage_arg & 0xffu) >= HK_STAGE_ANTIALIAS) rc = ctx_join_side(m->ctx[i]);  // (exchanges D / E read what the post stream wrote)
  // 1. "produced" on my stream
  if (!rc && hipEventRecord(m->produced[i], stream) != hipSuccess) { set_error("hipEventRecord failed (band %u)", i); rc = HK_E_HIP; }
  if (rc) P->failed.store(1, std::memory_order_release);
  if (!P->barrier()) return rc ? rc : HK_E_HIP;
  // 2. wait for the owners of the rows I receive, copy them in, "copied" on my stream
  std::vector<uint8_t> waited(n, 0);
  bool copies = false;
  for (const HkTransfer& t : *plan) {
    if (!t.is_recv || rc) continue;
    if (t.peer >= n) { set_error("bad peer in schedule"); rc = HK_E_INVALID; break; }
    if (!waited[t.peer]) {
      if (hipStreamWaitEvent(stream, m->produced[t.peer], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed (band %u)", i); rc = HK_E_HIP; break; }
      waited[t.peer] = 1;
    }
    size_t ldst = 0, lsrc = 0;
    char* dst = static_cast<char*>(ctx_buffer(m->ctx[i], t.buffer, &ldst));
    const char* src = static_cast<const char*>(ctx_buffer(m->ctx[t.peer], t.buffer, &lsrc));
    if (!dst || !src || t.offset + t.bytes > ldst || t.offset + t.bytes > lsrc) { set_error("halo transfer outside buffer %u", t.buffer); rc = HK_E_INVALID; break; }
    const hipError_t e = m->device[i] == m->device[t.peer] ? hipMemcpyAsync(dst + t.offset, src + t.offset, t.bytes, hipMemcpyDeviceToDevice, stream)
                                                             : hipMemcpyPeerAsync(dst + t.offset, m->device[i], src + t.offset, m->device[t.peer], t.bytes, stream);
    if (e != hipSuccess) { set_error("halo copy failed (band %u): %s", i, hipGetErrorString(e)); rc = HK_E_HIP; break; }
    P->bytes[i] += t.bytes;
    copies = true;
  }
  if (!rc && copies && hipEventRecord(m->copied[i], stream) != hipSuccess) { set_error("hipEventRecord failed (band %u)", i); rc = HK_E_HIP; }
  if (rc) P->failed.store(1, std::memory_order_release);
  if (!P->barrier()) return rc ? rc : HK_E_HIP;
  // 3. I do not run ahead of the copies that read my rows: every transfer I "send" names a band that copied from me
  std::fill(waited.begin(), waited.end(), 0);
  for (const HkTransfer& t : *plan) {
    if (t.is_recv || t.peer >= n || waited[t.peer]) continue;
    waited[t.peer] = 1;
    if (hipStreamWaitEvent(stream, m->copied[t.peer], 0) != hipSuccess) { set_error("hipStreamWaitEvent failed (band %u)", i); return HK_E_HIP; }
  }
  return HK_OK;
}

int band_frame(hk_multi* m, uint32_t i, const MultiJob& j) {
  hk_ctx* c = m->ctx[i];
  MultiPool* P = m->pool;
  if (hipSetDevice(m->device[i]) != hipSuccess) { set_error("hipSetDevice(%d) failed", m->device[i]); P->failed.store(1); return HK_E_HIP; }
  int rc = hk_frame_begin(c, j.f, j.v, j.pv, j.l);
  uint32_t hist = 0;  // the frame's history halo: every band derives the same count from the same uniforms (a failed band leaves 0,
  if (!rc) rc = hk_history_rows(c, &hist);  // and the others through the barrier's `failed`)
  hist <<= 8;
  auto step = [&](uint32_t s, bool exchange, uint32_t stage_arg) {
    if (exchange) {
      if (rc) P->failed.store(1, std::memory_order_release);
      const int e = band_exchange(m, i, stage_arg, j.st);  // (a failed band still enters: the others leave the barrier through `failed`)
      if (!rc) rc = e;
    }
    if (!rc) rc = hk_frame_stage(c, s, j.st, j.flags);
    if (rc) P->failed.store(1, std::memory_order_release);
  };
  for (uint32_t s = 0; s <= HK_STAGE_POST_PROCESS; ++s) step(s, s != HK_STAGE_TEMPORAL || hist, s <= HK_STAGE_SPATIAL ? (s | hist) : s);
  if (j.flags & HK_FRAME_ANTIALIAS) {
    step(HK_STAGE_ANTIALIAS, true, HK_STAGE_ANTIALIAS | hist);
    step(HK_STAGE_UPSCALE, j.st->upscale_kind == HK_UPSCALE_FSR1, HK_STAGE_UPSCALE);
  }
  return rc;
}

void pool_worker(hk_multi* m, uint32_t i) {
  MultiPool* P = m->pool;
  uint64_t seen = 0;
  for (;;) {
    MultiJob job;
    {
      std::unique_lock<std::mutex> lk(P->mu);
      P->cv_job.wait(lk, [&] { return P->quit || P->job_id != seen; });
      if (P->quit) return;
      seen = P->job_id;
      job = P->job;
    }
    set_error("%s", "");  // (this thread's error string: a band that fails only because another one did must not report a message of an earlier frame)
    const int rc = band_frame(m, i, job);
    {
      std::lock_guard<std::mutex> lk(P->mu);
      P->rc[i] = rc;
      P->err[i] = rc ? hk_last_error() : "";
      if (++P->done == P->n) P->cv_done.notify_one();
    }
  }
}

int pool_start(hk_multi* m) {
  if (m->pool) return HK_OK;
  MultiPool* P = new (std::nothrow) MultiPool();
  HK_REQUIRE(P, HK_E_NOMEM, "allocation failed");
  P->n = (uint32_t)m->ctx.size();
  P->rc.assign(P->n, 0);
  P->err.assign(P->n, "");
  P->cache.resize(P->n);
  P->bytes.assign(P->n, 0);
  m->pool = P;
  try {
    for (uint32_t i = 0; i < P->n; ++i) P->threads.emplace_back(pool_worker, m, i);
  } catch (...) {
    set_error("could not start the enqueue threads");
    return HK_E_NOMEM;  // (hk_multi_destroy joins what did start)
  }
  return HK_OK;
}

void pool_stop(hk_multi* m) {
  MultiPool* P = m->pool;
  if (!P) return;
  {
    std::lock_guard<std::mutex> lk(P->mu);
    P->quit = true;
  }
  P->cv_job.notify_all();
  for (std::thread& t : P->threads)
    if (t.joinable()) t.join();
  delete P;
  m->pool = nullptr;
}

}  // namespace

extern "C" {

int hk_comm_unique_id(uint8_t id[HK_COMM_ID_BYTES]) {
  HK_REQUIRE(id, HK_E_INVALID, "id is NULL");
  static_assert(sizeof(ncclUniqueId) == HK_COMM_ID_BYTES, "ncclUniqueId size");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded (%s)", dlerror() ? "dlopen failed" : "symbols missing");
  ncclUniqueId u;
  HK_NCCL(R, R->GetUniqueId(&u));
  memcpy(id, &u, HK_COMM_ID_BYTES);
  return HK_OK;
}

int hk_comm_available(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_REQUIRE(rccl(), HK_E_UNSUPPORTED, "librccl could not be loaded (or lacks a symbol the exchange needs)");
  CtxInfo ci;
  const int rc = ctx_info(c, &ci);
  if (rc) return rc;
  HK_HIP(hipSetDevice(ci.device));
  return HK_OK;
}

int hk_comm_init(hk_ctx* c, uint32_t rank, uint32_t n_ranks, const uint8_t id[HK_COMM_ID_BYTES]) {
  HK_REQUIRE(c && id && n_ranks > 0 && rank < n_ranks, HK_E_INVALID, "bad argument");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded");
  CtxInfo ci;
  int rc = ctx_info(c, &ci);
  if (rc) return rc;
  if (*ctx_comm_slot(c)) {  // exchanges of the previous communicator may still be enqueued on the context's stream
    if ((rc = hk_frame_wait(c))) return rc;
    comm_release(c);
  }
  HK_HIP(hipSetDevice(ci.device));
  Comm* cm = new (std::nothrow) Comm();
  HK_REQUIRE(cm, HK_E_NOMEM, "allocation failed");
  ncclUniqueId u;
  memcpy(&u, id, HK_COMM_ID_BYTES);
  ncclResult_t e = R->CommInitRank(&cm->comm, (int)n_ranks, u, (int)rank);
  if (e != ncclSuccess) {
    set_error("ncclCommInitRank(rank %u of %u, device %d) failed: %s", rank, n_ranks, ci.device, R->GetErrorString(e));
    delete cm;
    return HK_E_HIP;
  }
  cm->rank = rank;
  cm->n_ranks = n_ranks;
  if (hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&cm->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&cm->done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&cm->gather_done, hipEventDisableTiming) != hipSuccess) {
    set_error("cannot create the communicator's stream / events: %s", hipGetErrorString(hipGetLastError()));
    *ctx_comm_slot(c) = cm;  // (comm_release tears down what exists)
    comm_release(c);
    return HK_E_HIP;
  }
  // lane 1 (Comm): a second communicator over the same ranks, if this RCCL can split one; every rank takes the same branch (the symbol
  // is there or not in the one library a job runs), and a split that fails leaves lane 1 on lane 0
  if (R->CommSplit) {
    e = R->CommSplit(cm->comm, 0, (int)rank, &cm->comm2, nullptr);
    if (e != ncclSuccess || !cm->comm2) {
      cm->comm2 = nullptr;
    } else if (hipStreamCreateWithFlags(&cm->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&cm->ready2, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&cm->done2, hipEventDisableTiming) != hipSuccess) {
      set_error("cannot create the second communicator's stream / events: %s", hipGetErrorString(hipGetLastError()));
      *ctx_comm_slot(c) = cm;
      comm_release(c);
      return HK_E_HIP;
    }
  }
  *ctx_comm_slot(c) = cm;
  return hk_set_band(c, rank, n_ranks);
}

int hk_comm_destroy(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  const int rc = hk_frame_wait(c);
  comm_release(c);
  return rc;
}

int hk_comm_set_history_rows(hk_ctx* c, uint32_t rows) { return hk_set_history_rows(c, rows); }  // (the ABI 6 name)

int hk_comm_exchange(hk_ctx* c, uint32_t stage, const HkSettings* st) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  return comm_exchange(c, stage, st);
}

// hikari_hip_debug.h: rows [row_begin, row_end) of `src_buffer` travel to the same rows of `dst_buffer` of the SAME context as an
// ncclSend to the own rank paired with an ncclRecv from it - the group, the calls and the stream the halo exchanges use, on a box
// with one GPU (RCCL refuses two ranks on one device, so this is the one way the send / receive path can execute there)
int hk_debug_comm_loopback(hk_ctx* c, uint32_t src_buffer, uint32_t dst_buffer, uint32_t row_begin, uint32_t row_end, uint32_t mode) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  HK_REQUIRE(cm && cm->comm, HK_E_NOT_READY, "no communicator attached (hk_comm_init)");
  Rccl* R = rccl();
  HK_REQUIRE(R, HK_E_UNSUPPORTED, "librccl could not be loaded");
  CtxInfo ci;
  int rc = ctx_info(c, &ci);
  if (rc) return rc;
  uint32_t sw = 0, sh = 0, sb = 0, dw = 0, dh = 0, db = 0;
  if ((rc = hk_buffer_info(c, src_buffer, &sw, &sh, &sb)) || (rc = hk_buffer_info(c, dst_buffer, &dw, &dh, &db))) return rc;
  HK_REQUIRE(src_buffer != dst_buffer && sw == dw && sh == dh && sb == db && row_begin < row_end && row_end <= sh, HK_E_INVALID, "the two buffers must have one shape and the rows lie inside it");
  const uint64_t row_bytes = (uint64_t)sw * sb;
  HkTransfer tr[2] = {};
  tr[0].buffer = src_buffer; tr[0].peer = cm->rank; tr[0].is_recv = 0; tr[0].offset = row_begin * row_bytes; tr[0].bytes = (row_end - row_begin) * row_bytes;
  tr[1] = tr[0];
  tr[1].buffer = dst_buffer; tr[1].is_recv = 1;
  HK_HIP(hipSetDevice(ci.device));
  HK_REQUIRE(mode <= 2u, HK_E_INVALID, "mode 0, 1 or 2");
  const bool overlap = mode == 1u;  // 1: like the gather of a finished frame - nobody waits, comm_join does (hk_frame_begin of the same parity, any read)
  if (overlap && (rc = comm_join(c, -1))) return rc;  // (as comm_gather: never two overlapped transfers behind one record)
  if ((rc = run_ordered(c, cm, R, tr, 2, (hipStream_t)ci.stream, !overlap, (mode == 1u || mode == 2u) ? 1 : 0))) return rc;   // (mode 1 / 2: the gather's lane)
  if (overlap) {
    cm->gather_pending = true;
    cm->gather_parity = ci.frame_number & 1u;
  }
  cm->exchanges += 1;
  return HK_OK;
}

// hikari_hip_debug.h: 2 when the context's communicator has its second lane (a communicator and stream of its own for exchange B, the
// gather and exchanges D / E), 1 when the library could not split one off and those share the first lane
int hk_debug_comm_lanes(hk_ctx* c, uint32_t* lanes) {
  HK_REQUIRE(c && lanes, HK_E_INVALID, "bad argument");
  Comm* cm = static_cast<Comm*>(*ctx_comm_slot(c));
  HK_REQUIRE(cm && cm->comm, HK_E_NOT_READY, "no communicator attached (hk_comm_init)");
  *lanes = cm->comm2 && cm->stream2 ? 2u : 1u;
  return HK_OK;
}

int hk_comm_gather(hk_ctx* c, uint32_t buffer, uint32_t root) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  return comm_gather(c, buffer, root, false);   // by hand: complete in stream order, like the exchanges
}

// one process, n bands: the root band's context ends up holding the whole image on ITS device (peer copies on the root's stream,
// each ordered behind the owner's stream by an event); nothing waits on the host
int hk_multi_gather(hk_multi* m, uint32_t buffer, uint32_t root) {
  HK_REQUIRE(m && root < m->ctx.size(), HK_E_INVALID, "bad argument");
  const uint32_t n = (uint32_t)m->ctx.size();
  if (n < 2) return HK_OK;
  CtxInfo cr;
  int rc = ctx_info(m->ctx[root], &cr);
  if (rc) return rc;
  HK_REQUIRE(cr.width > 0, HK_E_NOT_READY, "hk_multi_resize has not been called");
  uint32_t nt = 0;
  if ((rc = hk_band_gather_schedule(cr.width, cr.height, cr.ratio, cr.upscale_kind, cr.band_bounds, root, n, root, buffer, nullptr, &nt))) return rc;
  std::vector<HkTransfer> tr(nt);
  if (nt && (rc = hk_band_gather_schedule(cr.width, cr.height, cr.ratio, cr.upscale_kind, cr.band_bounds, root, n, root, buffer, tr.data(), &nt))) return rc;
  size_t ldst = 0;
  char* dst = static_cast<char*>(ctx_buffer(m->ctx[root], buffer, &ldst));
  HK_REQUIRE(dst, HK_E_INVALID, "buffer %u is not allocated", buffer);
  for (const HkTransfer& t : tr) {
    HK_REQUIRE(t.is_recv && t.peer < n, HK_E_INVALID, "bad gather schedule");
    CtxInfo cp;
    if ((rc = ctx_info(m->ctx[t.peer], &cp))) return rc;
    size_t lsrc = 0;
    const char* src = static_cast<const char*>(ctx_buffer(m->ctx[t.peer], buffer, &lsrc));
    HK_REQUIRE(src && t.offset + t.bytes <= ldst && t.offset + t.bytes <= lsrc, HK_E_INVALID, "gather transfer outside buffer %u", buffer);
    HK_HIP(hipSetDevice(m->device[t.peer]));
    if ((rc = ctx_join_side(m->ctx[t.peer]))) return rc;  // (the image may have been finished on the owner's post stream)
    HK_HIP(hipEventRecord(m->produced[t.peer], (hipStream_t)cp.stream));
    HK_HIP(hipSetDevice(m->device[root]));
    HK_HIP(hipStreamWaitEvent((hipStream_t)cr.stream, m->produced[t.peer], 0));
    if (m->device[root] == m->device[t.peer])
      HK_HIP(hipMemcpyAsync(dst + t.offset, src + t.offset, t.bytes, hipMemcpyDeviceToDevice, (hipStream_t)cr.stream));
    else
      HK_HIP(hipMemcpyPeerAsync(dst + t.offset, m->device[root], src + t.offset, m->device[t.peer], t.bytes, (hipStream_t)cr.stream));
    m->bytes_copied += t.bytes;
  }
  // the owners do not overwrite their rows (next frame) before the root has copied them
  HK_HIP(hipEventRecord(m->copied[root], (hipStream_t)cr.stream));
  for (const HkTransfer& t : tr) {
    CtxInfo cp;
    if ((rc = ctx_info(m->ctx[t.peer], &cp))) return rc;
    HK_HIP(hipSetDevice(m->device[t.peer]));
    HK_HIP(hipStreamWaitEvent((hipStream_t)cp.stream, m->copied[root], 0));
  }
  return HK_OK;
}

int hk_multi_create(uint32_t n, const int* device_ids, uint32_t flags, hk_multi** out) {
  HK_REQUIRE(out && device_ids && n > 0 && n <= 64, HK_E_INVALID, "bad argument");
  *out = nullptr;
  hk_multi* m = new (std::nothrow) hk_multi();
  HK_REQUIRE(m, HK_E_NOMEM, "allocation failed");
  for (uint32_t i = 0; i < n; ++i) {
    hk_ctx* c = nullptr;
    int rc = hk_create(device_ids[i], flags, &c);
    if (!rc) rc = hk_set_band(c, i, n);
    if (rc) {
      if (c) hk_destroy(c);
      hk_multi_destroy(m);
      return rc;
    }
    m->ctx.push_back(c);
    m->device.push_back(device_ids[i]);
    hipEvent_t a = nullptr, b = nullptr;
    if (hipSetDevice(device_ids[i]) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
      set_error("event creation failed on device %d", device_ids[i]);
      if (a) (void)hipEventDestroy(a);
      hk_multi_destroy(m);
      return HK_E_HIP;
    }
    m->produced.push_back(a);
    m->copied.push_back(b);
  }
  // neighbouring bands copy from each other: direct xGMI access where the devices differ
  for (uint32_t i = 0; i + 1 < n; ++i) {
    const int a = device_ids[i], b = device_ids[i + 1];
    if (a == b) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
      (void)hipSetDevice(a);
      (void)hipDeviceEnablePeerAccess(b, 0);  // hipErrorPeerAccessAlreadyEnabled is fine
      (void)hipSetDevice(b);
      (void)hipDeviceEnablePeerAccess(a, 0);
      (void)hipGetLastError();
    }
  }
  *out = m;
  return HK_OK;
}

void hk_multi_destroy(hk_multi* m) {
  if (!m) return;
  pool_stop(m);
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    hk_destroy(m->ctx[i]);
    if (i < m->produced.size()) {
      (void)hipSetDevice(m->device[i]);
      (void)hipEventDestroy(m->produced[i]);
      (void)hipEventDestroy(m->copied[i]);
    }
  }
  delete m;
}

int hk_multi_context(hk_multi* m, uint32_t i, hk_ctx** out) {
  HK_REQUIRE(m && out && i < m->ctx.size(), HK_E_INVALID, "bad argument");
  *out = m->ctx[i];
  return HK_OK;
}

#define HK_EACH(call)                   \
  HK_REQUIRE(m, HK_E_INVALID, "multi is NULL"); \
  for (hk_ctx * c : m->ctx) {           \
    const int rc_ = (call);             \
    if (rc_) return rc_;                \
  }                                     \
  return HK_OK

int hk_multi_upload_scene(hk_multi* m, const hk_scene_builder* b) { HK_EACH(hk_upload_scene(c, b)); }
int hk_multi_upload_scene_instances(hk_multi* m, const hk_scene_builder* b) { HK_EACH(hk_upload_scene_instances(c, b)); }
int hk_multi_rebuild_scene_trees(hk_multi* m, uint32_t mode) { HK_EACH(hk_rebuild_scene_trees(c, mode)); }
// the same explicit split on every band's context (all of them must agree: the schedules are derived per context)
int hk_multi_set_band_bounds(hk_multi* m, const uint32_t* bounds, uint32_t n_bounds) { HK_EACH(hk_set_band_bounds(c, bounds, n_bounds)); }
// the rows of the history reservoirs that change owner travel between the bands' contexts, then every band takes the new split
int hk_multi_migrate_bands(hk_multi* m, const uint32_t* new_bounds, uint32_t n_bounds, uint32_t next_frame_number, const HkSettings* st) {
  HK_REQUIRE(m && st, HK_E_INVALID, "NULL argument");
  const uint32_t n = (uint32_t)m->ctx.size();
  HK_REQUIRE(!new_bounds || n_bounds == n + 1, HK_E_INVALID, "need band_count + 1 = %u boundaries", n + 1);
  int rc;
  if (n > 1) {
    std::vector<CtxInfo> ci(n);
    std::vector<std::vector<HkTransfer>> lists(n);
    std::vector<const std::vector<HkTransfer>*> plans(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
      if ((rc = ctx_join_side(m->ctx[i])) || (rc = ctx_info(m->ctx[i], &ci[i]))) return rc;
      HK_REQUIRE(ci[i].width > 0, HK_E_NOT_READY, "hk_multi_resize has not been called");
      uint32_t k = 0;
      if ((rc = hk_band_migration_schedule(ci[i].width, ci[i].height, ci[i].ratio, ci[i].band_bounds, new_bounds, i, n, next_frame_number, st, nullptr, &k))) return rc;
      lists[i].resize(k);
      if (k && (rc = hk_band_migration_schedule(ci[i].width, ci[i].height, ci[i].ratio, ci[i].band_bounds, new_bounds, i, n, next_frame_number, st, lists[i].data(), &k))) return rc;
      plans[i] = &lists[i];
    }
    if ((rc = multi_run_plans(m, plans, ci))) return rc;
  }
  return hk_multi_set_band_bounds(m, new_bounds, new_bounds ? n_bounds : 0u);
}
// the builder is finished ONCE (its transform bookkeeping advances once), every band's replica takes the records and builds its trees
int hk_multi_update_scene_instances(hk_multi* m, hk_scene_builder* b, uint32_t tree_mode) {
  HK_REQUIRE(m && b, HK_E_INVALID, "NULL argument");
  const int rc = hk_scene_builder_finish_instances(b);
  if (rc) return rc;
  uint32_t ni = 0;
  const HkInstance* inst = nullptr;
  const int rq = hk_scene_builder_instances(b, &inst, &ni);
  if (rq) return rq;
  for (hk_ctx* c : m->ctx) {
    int r = upload_scene_instances_unchecked(c, b);   // (stand-in trees: the device build follows)
    if (!r && ni >= 2) r = hk_rebuild_scene_trees(c, tree_mode);
    if (r) return r;
  }
  return HK_OK;
}
int hk_multi_upload_textures(hk_multi* m, const HkImageDesc* images, uint32_t n) { HK_EACH(hk_upload_textures(c, images, n)); }
int hk_multi_upload_noise(hk_multi* m, const uint8_t* rgba, size_t bytes) { HK_EACH(hk_upload_noise(c, rgba, bytes)); }
int hk_multi_resize(hk_multi* m, uint32_t w, uint32_t h, float ratio) {
  HK_REQUIRE(m, HK_E_INVALID, "multi is NULL");
  m->cache.clear();
  HK_EACH(hk_resize(c, w, h, ratio));
}
int hk_multi_wait(hk_multi* m) { HK_EACH(hk_frame_wait(c)); }
#undef HK_EACH
// every band's context refits its own replica of the scene; the builder's bookkeeping advances once
int hk_multi_refit_scene_instances(hk_multi* m, hk_scene_builder* b, uint32_t* moved) {
  HK_REQUIRE(m && b, HK_E_INVALID, "NULL argument");
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    const int rc = hk::refit_instances_impl(m->ctx[i], b, moved, i + 1 == m->ctx.size());
    if (rc) return rc;
  }
  return HK_OK;
}

// a count, or HK_HISTORY_AUTO (the default): every band's context derives the same halo from the frame's uniforms (hk_frame_begin)
int hk_multi_set_history_rows(hk_multi* m, uint32_t rows) {
  HK_REQUIRE(m && rows <= HK_HISTORY_AUTO, HK_E_INVALID, "bad argument");
  for (hk_ctx* c : m->ctx) {
    const int rc = hk_set_history_rows(c, rows);
    if (rc) return rc;
  }
  return HK_OK;
}

static int multi_frame_render_bands(hk_multi* m, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l, const HkSettings* st, uint32_t flags);
int hk_debug_multi_serial(int on) {
  g_multi_serial.store(on ? 1 : 0);
  return HK_OK;
}
int hk_multi_frame_render(hk_multi* m, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l, const HkSettings* st, uint32_t flags) {
  HK_REQUIRE(m && st, HK_E_INVALID, "bad argument");
  const int rc = multi_frame_render_bands(m, f, v, pv, l, st, flags);
  if (rc || !(flags & HK_FRAME_GATHER)) return rc;
  return hk_multi_gather(m, hk_final_buffer(st, flags), 0);  // band 0's device presents the image
}
static int multi_frame_render_bands(hk_multi* m, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l, const HkSettings* st, uint32_t flags) {
  int rc;
  {
    // every band's enqueue work on its own thread (the frame that re-splits the bands sets state on every context: serial)
    if (m->ctx.size() > 1 && !g_multi_serial.load() && !(flags & HK_FRAME_BALANCE_BANDS)) {
      if ((rc = pool_start(m))) return rc;
      MultiPool* P = m->pool;
      HK_REQUIRE(P->threads.size() == m->ctx.size(), HK_E_NOMEM, "the enqueue threads did not start");
      {
        std::unique_lock<std::mutex> lk(P->mu);
        P->job = MultiJob{f, v, pv, l, st, flags};
        P->done = 0;
        P->failed.store(0);
        P->arrived.store(0);  // (a failed frame leaves its barriers half-entered; every thread is idle here)
        P->job_id += 1;
        P->cv_job.notify_all();
        P->cv_done.wait(lk, [&] { return P->done == P->n; });
      }
      for (uint32_t i = 0; i < P->n; ++i) {
        m->bytes_copied += P->bytes[i];
        P->bytes[i] = 0;
      }
      for (uint32_t i = 0; i < P->n; ++i)
        if (P->rc[i] && !P->err[i].empty()) {  // the band that failed first-hand (the others report "some band failed")
          set_error("band %u: %s", i, P->err[i].c_str());
          return P->rc[i];
        }
      for (uint32_t i = 0; i < P->n; ++i)
        if (P->rc[i]) return P->rc[i];
      return HK_OK;
    }
  }
  for (hk_ctx* c : m->ctx)
    if ((rc = hk_frame_begin(c, f, v, pv, l))) return rc;
  if ((flags & HK_FRAME_BALANCE_BANDS) && m->ctx.size() > 1) {  // one context counts (they all hold the same G-buffer), all take the split
    std::vector<uint32_t> bounds(m->ctx.size() + 1);
    if ((rc = hk_set_view_options(m->ctx[0], st->taa, st->upscale_kind, st->upscale_sharpness))) return rc;  // (jitter of the primary rays)
    if ((rc = hk_balance_bands(m->ctx[0], 0, bounds.data(), (uint32_t)bounds.size()))) return rc;
    if ((rc = hk_multi_set_band_bounds(m, bounds.data(), (uint32_t)bounds.size()))) return rc;
  }
  uint32_t hist = 0;
  if ((rc = hk_history_rows(m->ctx[0], &hist))) return rc;  // (the same on every band: derived from the same uniforms)
  hist <<= 8;
  auto stage = [&](uint32_t s) {
    for (hk_ctx* c : m->ctx) {
      const int r = hk_frame_stage(c, s, st, flags);
      if (r) return r;
    }
    return (int)HK_OK;
  };
  for (uint32_t s = 0; s <= HK_STAGE_POST_PROCESS; ++s) {
    if ((s != HK_STAGE_TEMPORAL || hist) && (rc = multi_exchange(m, s <= HK_STAGE_SPATIAL ? (s | hist) : s, st))) return rc;
    if ((rc = stage(s))) return rc;
  }
  if (flags & HK_FRAME_ANTIALIAS) {
    if ((rc = multi_exchange(m, HK_STAGE_ANTIALIAS | hist, st))) return rc;
    if ((rc = stage(HK_STAGE_ANTIALIAS))) return rc;
    if (st->upscale_kind == HK_UPSCALE_FSR1 && (rc = multi_exchange(m, HK_STAGE_UPSCALE, st))) return rc;
    if ((rc = stage(HK_STAGE_UPSCALE))) return rc;
  }
  return HK_OK;
}

int hk_multi_read_buffer(hk_multi* m, uint32_t buffer, void* dst, size_t bytes) {
  HK_REQUIRE(m && dst && !m->ctx.empty(), HK_E_INVALID, "bad argument");
  const uint32_t n = (uint32_t)m->ctx.size();
  uint32_t bw = 0, bh = 0, bpp = 0;
  int rc = hk_buffer_info(m->ctx[0], buffer, &bw, &bh, &bpp);
  if (rc) return rc;
  CtxInfo ci;
  if ((rc = ctx_info(m->ctx[0], &ci))) return rc;
  uint32_t rw, rh;
  if ((rc = hk_scaled_size(ci.width, ci.height, ci.ratio, &rw, &rh))) return rc;
  // reservoir buffers are allocated window-size but indexed as rw x rh records (light.rs:344, light.wgsl:1061)
  const bool reservoir = buffer >= HK_BUF_RESERVOIR0 && buffer < HK_BUF_RESERVOIR0 + 10;
  const uint32_t rows = reservoir ? rh : bh;
  const size_t row_bytes = (size_t)(reservoir ? rw : bw) * bpp;
  HK_REQUIRE(bytes == (size_t)bw * bh * bpp, HK_E_INVALID, "size mismatch: buffer has %zu bytes", (size_t)bw * bh * bpp);
  // Which rows of `buffer` does band i own?  Bands partition the RENDER rows [b0, b1):
  //   render-size buffers (and the rw x rh records of a reservoir buffer)   [b0, b1)
  //   SMAA Tu4x outputs (upscale_output / taa_output, two rows per render row) [2 b0, 2 b1) clamped, the last band to the end
  //   FSR1 window-size outputs                                               hk_band_rows(height) (HK_STAGE_UPSCALE)
  //   full-size planes (G-buffer, albedo) at ratio > 1                        the window rows under [b0, b1) (every band ray-casts
  //                                                                           a superset of them, full_rows_for in context.hip)
  const bool upscaled = buffer == HK_BUF_UPSCALE_OUTPUT || buffer == HK_BUF_TAA_OUTPUT || buffer == HK_BUF_PREVIOUS_TAA_OUTPUT;
  const bool fsr_window = ci.upscale_kind == HK_UPSCALE_FSR1 && (buffer == HK_BUF_UPSCALE_OUTPUT || buffer == HK_BUF_UPSCALE_SHARPENED);
  std::vector<uint8_t> tmp(bytes);
  memset(dst, 0, bytes);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t b0, b1, y0, y1;
    band_rows_in(ci.band_bounds, rh, rh, i, n, &b0, &b1);
    if (fsr_window) {
      band_rows_in(ci.band_bounds, rh, ci.height, i, n, &y0, &y1);
    } else if (rows == rh) {
      y0 = b0; y1 = b1;
    } else if (upscaled && ci.upscale_kind == HK_UPSCALE_SMAA_TU4X) {
      y0 = 2 * b0 < rows ? 2 * b0 : rows;
      y1 = (b1 == rh) ? rows : (2 * b1 < rows ? 2 * b1 : rows);
    } else {
      y0 = (uint32_t)((uint64_t)b0 * rows / rh);
      y1 = (b1 == rh) ? rows : (uint32_t)((uint64_t)b1 * rows / rh);
    }
    if (y1 <= y0) continue;
    if ((rc = hk_read_buffer(m->ctx[i], buffer, tmp.data(), bytes))) return rc;
    memcpy(static_cast<uint8_t*>(dst) + (size_t)y0 * row_bytes, tmp.data() + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes);
  }
  return HK_OK;
}

}  // extern "C"
