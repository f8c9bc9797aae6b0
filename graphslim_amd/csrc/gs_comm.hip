// Rank-to-rank exchange of the sharded GCond iteration (include/gslim.h "multi-GPU"): everything that crosses ranks is an
// all-reduce(sum) of a contiguous engine buffer, issued by the library itself on the engine's stream, so that a rank makes ONE
// host call per outer iteration (gs_gcond_outer_step_sharded, gs_engine.hip).
//
// Three transports behind one handle:
//   rccl      ncclAllReduce on a communicator this library creates with ncclCommInitRank (one process per GPU, xGMI).  RCCL is
//             bound with dlopen / dlsym — the copy PyTorch-ROCm has already loaded if there is one — so that libgslim_hip.so has
//             no link-time dependency on it and loads on machines without RCCL (single-GPU use never touches this file's RCCL
//             half).  The declarations below restate the few entry points of rccl.h (ROCm 7.2: /opt/rocm/include/rccl/rccl.h:40-43
//             ncclUniqueId, :187 ncclGetUniqueId, :220 ncclCommInitRank, :611 ncclAllReduce).
//   callback  the caller's function performs the exchange (tests: torch.distributed over gloo between processes that share one
//             GPU — RCCL refuses several ranks per device — and an in-process sum between virtual ranks).
//   null      no exchange (world 1, and timing-only virtual ranks: tools/virtual_scale.py).
// The reference has no distributed code: nothing here replaces a reference line.
#include <dlfcn.h>

#include <new>

#include "gs_common.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // rccl.h:40-43 (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
// rccl.h enums: ncclDataType_t ncclFloat32 = 7, ncclFloat64 = 8; ncclRedOp_t ncclSum = 0; ncclResult_t ncclSuccess = 0
enum { RCCL_F32 = 7, RCCL_F64 = 8, RCCL_SUM = 0 };

struct rccl_api {
  void* handle;
  int (*GetUniqueId)(rccl_unique_id*);
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int);
  int (*CommDestroy)(rccl_comm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  int (*GetVersion)(int*);
};

rccl_api g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int load_rccl() {
  if (g_rccl.handle) return GS_OK;
  // the copy already in the process first (PyTorch-ROCm ships its own librccl.so and has loaded it when torch.distributed is
  // in use: two RCCL copies in one process would each run their own bootstrap / proxy threads)
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h)
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) GS_FAIL(GS_ERR_STATE, "gs_comm: librccl.so not found (%s)", dlerror());
  rccl_api a;
  a.handle = h;
  a.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  a.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");      // rccl.h: ncclGetVersion(int* version) — optional, allocates nothing
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce)
    GS_FAIL(GS_ERR_STATE, "gs_comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
  g_rccl = a;
  return GS_OK;
}

const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"; }

}  // namespace

enum { GS_COMM_NULL = 0, GS_COMM_RCCL = 1, GS_COMM_CALLBACK = 2 };
#define GS_COMM_TAGS 8
#define GS_COMM_MAX_EVENTS 8192

struct gs_comm {
  int kind, rank, world;
  rccl_comm_t nccl;
  gs_allreduce_fn fn;
  void* user;
  // bookkeeping per exchange point (tag): calls and bytes always; event pairs on the stream while timing is on
  long calls[GS_COMM_TAGS], bytes[GS_COMM_TAGS];
  bool timing;
  int n_ev;
  hipEvent_t* ev;        // [2 * GS_COMM_MAX_EVENTS], created lazily
  int* ev_tag;
};

// Is the RCCL binding usable in this process?  dlopen + the entry points + ncclGetVersion: allocates nothing and starts no bootstrap
// thread (ncclGetUniqueId does: ADVICE r04) — what every rank checks before any of them enters the collective ncclCommInitRank.
extern "C" int gs_comm_probe(int* rccl_version) {
  GS_TRY(load_rccl());
  int v = 0;
  if (g_rccl.GetVersion) {
    const int rc = g_rccl.GetVersion(&v);
    if (rc != 0) GS_FAIL(GS_ERR_STATE, "ncclGetVersion failed: %s", rccl_err(rc));
  }
  if (rccl_version) *rccl_version = v;
  return GS_OK;
}

extern "C" int gs_comm_unique_id(void* id128) {
  if (!id128) GS_FAIL(GS_ERR_ARG, "gs_comm_unique_id: null argument");
  GS_TRY(load_rccl());
  rccl_unique_id id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) GS_FAIL(GS_ERR_STATE, "ncclGetUniqueId failed: %s", rccl_err(rc));
  memcpy(id128, &id, sizeof(id));
  return GS_OK;
}

static gs_comm* new_comm(int kind, int rank, int world) {
  gs_comm* c = new (std::nothrow) gs_comm;
  if (!c) return nullptr;
  memset(c, 0, sizeof(*c));
  c->kind = kind; c->rank = rank; c->world = world;
  return c;
}

extern "C" int gs_comm_create_rccl(const void* id128, int rank, int world, gs_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) GS_FAIL(GS_ERR_ARG, "gs_comm_create_rccl: bad argument");
  GS_TRY(load_rccl());
  gs_comm* c = new_comm(GS_COMM_RCCL, rank, world);
  if (!c) GS_FAIL(GS_ERR_STATE, "gs_comm_create_rccl: out of host memory");
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  const int rc = g_rccl.CommInitRank(&c->nccl, world, id, rank);      // on the calling thread's current device
  if (rc != 0) {
    delete c;
    GS_FAIL(GS_ERR_STATE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_err(rc));
  }
  *out = c;
  return GS_OK;
}

extern "C" int gs_comm_create_callback(gs_allreduce_fn fn, void* user, int rank, int world, gs_comm** out) {
  if (!fn || !out || world < 1 || rank < 0 || rank >= world) GS_FAIL(GS_ERR_ARG, "gs_comm_create_callback: bad argument");
  gs_comm* c = new_comm(GS_COMM_CALLBACK, rank, world);
  if (!c) GS_FAIL(GS_ERR_STATE, "gs_comm_create_callback: out of host memory");
  c->fn = fn; c->user = user;
  *out = c;
  return GS_OK;
}

extern "C" int gs_comm_create_null(int rank, int world, gs_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world) GS_FAIL(GS_ERR_ARG, "gs_comm_create_null: bad argument");
  gs_comm* c = new_comm(GS_COMM_NULL, rank, world);
  if (!c) GS_FAIL(GS_ERR_STATE, "gs_comm_create_null: out of host memory");
  *out = c;
  return GS_OK;
}

extern "C" void gs_comm_destroy(gs_comm* c) {
  if (!c) return;
  if (c->ev) {
    for (int i = 0; i < 2 * GS_COMM_MAX_EVENTS; ++i)
      if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete[] c->ev;
    delete[] c->ev_tag;
  }
  if (c->kind == GS_COMM_RCCL && c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
  delete c;
}

extern "C" int gs_comm_rank(const gs_comm* c) { return c ? c->rank : -1; }
extern "C" int gs_comm_world(const gs_comm* c) { return c ? c->world : -1; }

extern "C" int gs_comm_set_timing(gs_comm* c, int on) {
  if (!c) GS_FAIL(GS_ERR_ARG, "gs_comm_set_timing: null communicator");
  c->timing = on != 0;
  if (on) {
    c->n_ev = 0;
    for (int t = 0; t < GS_COMM_TAGS; ++t) c->calls[t] = c->bytes[t] = 0;
    if (!c->ev) {
      c->ev = new (std::nothrow) hipEvent_t[2 * GS_COMM_MAX_EVENTS];
      c->ev_tag = new (std::nothrow) int[GS_COMM_MAX_EVENTS];
      if (!c->ev || !c->ev_tag) GS_FAIL(GS_ERR_STATE, "gs_comm_set_timing: out of host memory");
      for (int i = 0; i < 2 * GS_COMM_MAX_EVENTS; ++i) c->ev[i] = nullptr;
    }
  }
  return GS_OK;
}

// calls / bytes / summed microseconds per exchange point since timing was switched on (arrays of GS_COMM_TAGS = 8; tags:
// 0 BN-2 sums, 1 pair scores, 2 [dX~ | d adj | loss_c], 3 BN-2 backward sums, 4 marginals (+ dW1 | db1), 5 other).
// Synchronises the recorded events.
// timed_calls (optional): per exchange point the number of calls total_us covers — the event pool holds GS_COMM_MAX_EVENTS pairs, calls
// beyond it are counted in `calls` / `bytes` but not timed (dividing total_us by `calls` under-reported long windows)
extern "C" int gs_comm_read_timing2(gs_comm* c, long* calls, long* bytes, double* total_us, long* timed_calls) {
  if (!c || !calls || !bytes || !total_us) GS_FAIL(GS_ERR_ARG, "gs_comm_read_timing: null argument");
  for (int t = 0; t < GS_COMM_TAGS; ++t) {
    calls[t] = c->calls[t]; bytes[t] = c->bytes[t]; total_us[t] = 0.0;
    if (timed_calls) timed_calls[t] = 0;
  }
  for (int i = 0; i < c->n_ev; ++i) {
    float ms = 0.f;
    GS_HIP(hipEventSynchronize(c->ev[2 * i + 1]));
    GS_HIP(hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
    total_us[c->ev_tag[i]] += 1e3 * (double)ms;
    if (timed_calls) timed_calls[c->ev_tag[i]] += 1;
  }
  return GS_OK;
}
extern "C" int gs_comm_read_timing(gs_comm* c, long* calls, long* bytes, double* total_us) {
  return gs_comm_read_timing2(c, calls, bytes, total_us, nullptr);
}

// all-reduce(sum) in place on `stream`.  dtype: 0 = fp32, 1 = fp64.  tag: exchange point (bookkeeping only).
extern "C" int gs_comm_allreduce(gs_comm* c, void* buf, long count, int dtype, int tag, gs_stream_t stream) {
  if (!c || (!buf && count > 0) || count < 0 || (dtype != 0 && dtype != 1)) GS_FAIL(GS_ERR_ARG, "gs_comm_allreduce: bad argument");
  if (tag < 0 || tag >= GS_COMM_TAGS) tag = GS_COMM_TAGS - 1;
  hipStream_t s = (hipStream_t)stream;
  c->calls[tag] += 1;
  c->bytes[tag] += count * (dtype ? 8 : 4);
  if (c->world == 1 && c->kind != GS_COMM_RCCL) return GS_OK;      // (a world-1 RCCL communicator still runs the call: 1-GPU test leg)
  if (count == 0) return GS_OK;
  const bool tm = c->timing && c->ev && c->n_ev < GS_COMM_MAX_EVENTS;
  if (tm) {
    for (int q = 0; q < 2; ++q)
      if (!c->ev[2 * c->n_ev + q]) GS_HIP(hipEventCreate(&c->ev[2 * c->n_ev + q]));
    GS_HIP(hipEventRecord(c->ev[2 * c->n_ev], s));
  }
  if (c->kind == GS_COMM_RCCL) {
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, dtype ? RCCL_F64 : RCCL_F32, RCCL_SUM, c->nccl, s);
    if (rc != 0) GS_FAIL(GS_ERR_STATE, "ncclAllReduce(%ld x %s) failed: %s", count, dtype ? "f64" : "f32", rccl_err(rc));
  } else if (c->kind == GS_COMM_CALLBACK) {
    const int rc = c->fn(c->user, buf, count, dtype, stream);
    if (rc != 0) GS_FAIL(GS_ERR_STATE, "gs_comm: the caller's all-reduce callback failed (%d)", rc);
  }
  if (tm) {
    GS_HIP(hipEventRecord(c->ev[2 * c->n_ev + 1], s));
    c->ev_tag[c->n_ev] = tag;
    c->n_ev++;
  }
  return GS_OK;
}
