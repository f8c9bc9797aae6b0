// Rank-to-rank exchange of the sharded GCond iteration (include/gslim.h "multi-GPU"): everything that crosses ranks is an
// all-reduce(sum) of a contiguous engine buffer, issued by the library itself on the engine's stream, so that a rank makes ONE
// host call per outer iteration (gs_gcond_outer_step_sharded, gs_engine.hip).
//
// Four transports behind one handle:
//   rccl      ncclAllReduce on a communicator this library creates with ncclCommInitRank (one process per GPU, xGMI).  RCCL is
//             bound with dlopen / dlsym — the copy PyTorch-ROCm has already loaded if there is one — so that libgslim_hip.so has
//             no link-time dependency on it and loads on machines without RCCL (single-GPU use never touches this file's RCCL
//             half).  The declarations below restate the few entry points of rccl.h (ROCm 7.2: /opt/rocm/include/rccl/rccl.h:40-43
//             ncclUniqueId, :187 ncclGetUniqueId, :220 ncclCommInitRank, :611 ncclAllReduce).
//   callback  the caller's function performs the exchange (tests: torch.distributed over gloo between processes that share one
//             GPU — RCCL refuses several ranks per device — and an in-process sum between virtual ranks).
//   ipc       (round 5) one-shot PEER-READ all-reduce for the sub-MB joins of the iteration: every rank owns an arena of device memory
//             (a flag, an arrival counter, two data slots), exports it with hipIpcGetMemHandle and maps every other rank's arena
//             (hipIpcOpenMemHandle: xGMI peer access between the GPUs of a node).  One launch per exchange: copy the contribution into
//             the own slot, publish the flag, wait for every peer's flag, read every peer's slot over the fabric and add them IN RANK
//             ORDER (so that all replicas end with bitwise the same sums), write the result in place.  No ring, no proxy thread, no
//             host round trip: SURVEY section 8e asks for exactly this for messages of <= 1.2 MB.  Larger exchanges go to a fallback
//             communicator (RCCL or callback) when one is given.
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

enum { GS_COMM_NULL = 0, GS_COMM_RCCL = 1, GS_COMM_CALLBACK = 2, GS_COMM_IPC = 3 };
#define GS_IPC_MAX_WORLD 16
#define GS_IPC_HEADER 4096      /* bytes in front of the data slots: [0] flag (u64), [64] arrival counter (u32), [128] error word (u32) */
#define GS_IPC_GRID 32          /* workgroups per exchange: resident whatever else runs (a waiting workgroup must never keep a peer's writer off the chip) */
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
  // ipc transport
  char* arena[GS_IPC_MAX_WORLD];      // every rank's arena as mapped HERE (own: the allocation itself)
  size_t slot_bytes;                  // capacity of one data slot
  unsigned long long seq;             // exchanges issued so far
  gs_comm* fallback;                  // exchanges beyond slot_bytes (not owned)
  bool own_arena;
  unsigned* err_host;                 // host-pinned, device-mapped error word: a timed-out flag wait writes rank + 1 here (sticky) and the
  unsigned* err_dev;                  // NEXT gs_comm_allreduce / gs_comm_ipc_error on the host sees it without a device synchronisation
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

// ---- ipc transport ---------------------------------------------------------------------------------------------------------------
struct gs_ipc_args {
  char* arena[GS_IPC_MAX_WORLD];
  int rank, world;
  unsigned long long seq;      // this exchange (1-based)
  size_t slot_bytes;
  void* buf; long count;       // in place; count elements of 4 (fp32) or 8 (fp64) bytes
  int f64;
  unsigned* err;               // host-visible sticky error word (gs_comm.err_dev)
};

template <typename T>
__device__ __forceinline__ void ipc_reduce(const gs_ipc_args& a, int par) {
  typedef T vec4 __attribute__((ext_vector_type(16 / sizeof(T))));
  constexpr int VE = 16 / sizeof(T);
  const long nv = a.count / VE;
  T* buf = (T*)a.buf;
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool aligned = (((uintptr_t)a.buf) & 15) == 0;      // (the slots are 16-byte aligned by construction; the caller's buffer need not be)
  for (long i = t0; i < nv; i += stride) {
    vec4 acc = *reinterpret_cast<const vec4*>(a.arena[0] + GS_IPC_HEADER + par * a.slot_bytes + i * 16);
    for (int r = 1; r < a.world; ++r) acc += *reinterpret_cast<const vec4*>(a.arena[r] + GS_IPC_HEADER + par * a.slot_bytes + i * 16);
    if (aligned) {
      *reinterpret_cast<vec4*>(buf + i * VE) = acc;
    } else {
#pragma unroll
      for (int q = 0; q < VE; ++q) buf[i * VE + q] = acc[q];
    }
  }
  for (long i = nv * VE + t0; i < a.count; i += stride) {
    T acc = *reinterpret_cast<const T*>(a.arena[0] + GS_IPC_HEADER + par * a.slot_bytes + i * sizeof(T));
    for (int r = 1; r < a.world; ++r) acc += *reinterpret_cast<const T*>(a.arena[r] + GS_IPC_HEADER + par * a.slot_bytes + i * sizeof(T));
    buf[i] = acc;
  }
}

__global__ __launch_bounds__(256) void gs_ipc_allreduce_kernel(gs_ipc_args a) {
  const int par = (int)(a.seq & 1);
  char* mine = a.arena[a.rank];
  // 1. my contribution -> my slot (the slot of this parity was last read in exchange seq - 2: every peer has since published seq - 1)
  {
    // (a buffer that is not 16-byte aligned goes through the 4-byte path as a whole: the transport is chosen from the COUNT alone, so that
    // every rank provably takes the same branch whatever its pointer — ADVICE r05)
    const long nbytes = a.count * (a.f64 ? 8 : 4), nb16 = (((uintptr_t)a.buf) & 15) ? 0 : nbytes / 16, tail0 = nb16 * 16;
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (long i = t0; i < nb16; i += stride)
      *reinterpret_cast<i32x4*>(mine + GS_IPC_HEADER + par * a.slot_bytes + i * 16) = *reinterpret_cast<const i32x4*>((const char*)a.buf + i * 16);
    for (long i = tail0 + 4 * t0; i < nbytes; i += 4 * stride)
      *reinterpret_cast<int*>(mine + GS_IPC_HEADER + par * a.slot_bytes + i) = *reinterpret_cast<const int*>((const char*)a.buf + i);
  }
  __threadfence_system();      // my stores are visible to the other devices before the flag below can be
  __syncthreads();
  // 2. the last workgroup to arrive publishes the flag (arrival counter: monotonic, gridDim.x per exchange)
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(mine + 64), 1u);
    if (old + 1u == (unsigned)(a.seq * gridDim.x))
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. wait for every rank's flag (bounded: ~2 s; a rank that never arrives must not hang the device).  A timeout is FATAL for the
  //    communicator: the error word — in the arena and in host-pinned memory — is sticky, this exchange leaves `buf`
  //    untouched (no silently wrong sums), and the host refuses every later exchange (gs_comm_allreduce returns GS_ERR_STATE).
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < a.world) {
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(a.arena[threadIdx.x]);
    long spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > 4000000L) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 128), 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.err) __hip_atomic_store(a.err, 1u + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        timed_out = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (timed_out) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: nothing read below may come from a line cached before the flags
  // 4. the sum, ranks in ascending order on every rank: the replicas end with bitwise the same values
  if (a.f64) ipc_reduce<double>(a, par);
  else ipc_reduce<float>(a, par);
}

// This rank's arena: device memory on the current device (uncached where the runtime offers it: the slots are written once and read
// once by other devices), zeroed; `handle64` receives the 64 bytes every other rank opens it with (hipIpcMemHandle_t).
extern "C" int gs_comm_ipc_arena(size_t slot_bytes, void** arena, void* handle64) {
  if (!arena || !handle64 || slot_bytes < 16) GS_FAIL(GS_ERR_ARG, "gs_comm_ipc_arena: bad argument");
  slot_bytes = (slot_bytes + 255) & ~(size_t)255;
  const size_t total = GS_IPC_HEADER + 2 * slot_bytes;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      GS_HIP(hipMalloc(&p, total));
    }
  }
  GS_HIP(hipMemset(p, 0, total));
  GS_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  const hipError_t rc = hipIpcGetMemHandle(&h, p);
  if (rc != hipSuccess) {
    (void)hipFree(p);
    GS_FAIL(GS_ERR_STATE, "gs_comm_ipc_arena: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on hosts with dmabuf IPC only)",
            hipGetErrorString(rc));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle64, &h, 64);
  *arena = p;
  return GS_OK;
}

// handles: [world][64] — every rank's handle in rank order (this rank's own entry is ignored: `arena` is used).  The communicator
// takes ownership of `arena`.  fallback (optional, not owned): carries the exchanges that do not fit a slot.
extern "C" int gs_comm_create_ipc(void* arena, const void* handles, size_t slot_bytes, int rank, int world, gs_comm* fallback, gs_comm** out) {
  if (!arena || !handles || !out || world < 1 || world > GS_IPC_MAX_WORLD || rank < 0 || rank >= world || slot_bytes < 16)
    GS_FAIL(GS_ERR_ARG, "gs_comm_create_ipc: bad argument (world <= %d)", GS_IPC_MAX_WORLD);
  gs_comm* c = new_comm(GS_COMM_IPC, rank, world);
  if (!c) GS_FAIL(GS_ERR_STATE, "gs_comm_create_ipc: out of host memory");
  c->slot_bytes = (slot_bytes + 255) & ~(size_t)255;
  c->fallback = fallback;
  c->own_arena = true;
  for (int r = 0; r < world; ++r) {
    if (r == rank) { c->arena[r] = (char*)arena; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + 64 * r, 64);
    void* p = nullptr;
    const hipError_t rc = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (rc != hipSuccess) {
      for (int q = 0; q < r; ++q)
        if (q != rank && c->arena[q]) (void)hipIpcCloseMemHandle(c->arena[q]);
      (void)hipFree(arena);
      delete c;
      GS_FAIL(GS_ERR_STATE, "gs_comm_create_ipc: hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(rc));
    }
    c->arena[r] = (char*)p;
  }
  // the sticky error word the host polls without synchronising (absent => gs_comm_ipc_error falls back to a device read)
  void* eh = nullptr;
  if (hipHostMalloc(&eh, 64, hipHostMallocMapped) == hipSuccess) {
    memset(eh, 0, 64);
    void* ed = nullptr;
    if (hipHostGetDevicePointer(&ed, eh, 0) == hipSuccess) { c->err_host = (unsigned*)eh; c->err_dev = (unsigned*)ed; }
    else { (void)hipGetLastError(); (void)hipHostFree(eh); }
  } else {
    (void)hipGetLastError();
  }
  *out = c;
  return GS_OK;
}

// 0 = every exchange so far found all its peers; r + 1 = a wait for rank r's flag timed out: that exchange left its buffer untouched and
// the communicator is dead (every later gs_comm_allreduce returns GS_ERR_STATE).  sync != 0 waits for the device first (exact for
// everything issued so far); 0 reads the host-visible word as it stands (what the product path polls once per epoch / read_loss).
static int ipc_error_word(gs_comm* c, bool sync, unsigned* v) {
  *v = 0;
  if (sync || !c->err_host) {
    GS_HIP(hipMemcpy(v, c->arena[c->rank] + 128, 4, hipMemcpyDeviceToHost));
  } else {
    *v = __atomic_load_n(c->err_host, __ATOMIC_ACQUIRE);
  }
  return GS_OK;
}
extern "C" int gs_comm_ipc_error(gs_comm* c, int* code) {
  if (!c || !code || c->kind != GS_COMM_IPC) GS_FAIL(GS_ERR_ARG, "gs_comm_ipc_error: not an ipc communicator");
  unsigned v = 0;
  GS_TRY(ipc_error_word(c, true, &v));
  *code = (int)v;
  return GS_OK;
}
// the product path's check (graphslim_amd.parallel.ShardedGCond.read_loss, once per epoch): GS_OK, or GS_ERR_STATE with the peer named.
// Communicators of other kinds have nothing to report.
extern "C" int gs_comm_check(gs_comm* c) {
  if (!c) GS_FAIL(GS_ERR_ARG, "gs_comm_check: null communicator");
  if (c->kind != GS_COMM_IPC) return c->fallback ? gs_comm_check(c->fallback) : GS_OK;
  unsigned v = 0;
  GS_TRY(ipc_error_word(c, false, &v));
  if (v) GS_FAIL(GS_ERR_STATE, "gs_comm (ipc): rank %d waited ~2 s for rank %u's flag and gave up: the exchange left its buffer untouched, "
                               "the replicas have diverged — the communicator is dead (use the rccl transport if ranks can be seconds apart)",
                 c->rank, v - 1u);
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
  if (c->kind == GS_COMM_IPC) {
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
      if (r != c->rank && c->arena[r]) (void)hipIpcCloseMemHandle(c->arena[r]);
    if (c->own_arena && c->arena[c->rank]) (void)hipFree(c->arena[c->rank]);
    if (c->err_host) (void)hipHostFree(c->err_host);
  }
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
  } else if (c->kind == GS_COMM_IPC) {
    const size_t nbytes = (size_t)count * (dtype ? 8 : 4);
    GS_TRY(gs_comm_check(c));      // a flag wait of an earlier exchange timed out: refuse (sticky)
    if (nbytes > c->slot_bytes) {
      // (decided from the count ALONE — the same on every rank — so every rank takes this branch together and `seq` stays in step; an
      // unaligned buffer goes through the kernel's 4-byte path)
      if (!c->fallback) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_comm (ipc): an exchange of %zu bytes does not fit the %zu-byte slots and no fallback "
                                                    "communicator was given", nbytes, c->slot_bytes);
      c->calls[tag] -= 1; c->bytes[tag] -= (long)nbytes;      // (counted by this handle above; the fallback does not double-count for us)
      const int rc = gs_comm_allreduce(c->fallback, buf, count, dtype, tag, stream);
      c->calls[tag] += 1; c->bytes[tag] += (long)nbytes;
      if (rc != GS_OK) return rc;
    } else {
      gs_ipc_args a;
      for (int r = 0; r < GS_IPC_MAX_WORLD; ++r) a.arena[r] = r < c->world ? c->arena[r] : nullptr;
      a.rank = c->rank; a.world = c->world; a.seq = ++c->seq; a.slot_bytes = c->slot_bytes; a.buf = buf; a.count = count; a.f64 = dtype; a.err = c->err_dev;
      hipLaunchKernelGGL(gs_ipc_allreduce_kernel, dim3(GS_IPC_GRID), dim3(256), 0, s, a);
      GS_LAUNCH_CHECK();
    }
  }
  if (tm) {
    GS_HIP(hipEventRecord(c->ev[2 * c->n_ev + 1], s));
    c->ev_tag[c->n_ev] = tag;
    c->n_ev++;
  }
  return GS_OK;
}
