// Multi-GPU candidate sharding inside the library: no PyTorch, no launcher.
//
// The reference evaluates every candidate of a random-search acquisition in ONE NumPy array in ONE
// process and takes obj_vals.argmax() (dragonfly/utils/oper_utils.py:59-80, :73).  Candidates are
// independent given the fitted GP, so they shard contiguously over the GPUs of a node; the fit is
// replicated (bit-identical on every device) and the only exchange is an RCCL all-gather of one
// 16-byte (value:f64, global index:i64) pair per rank over xGMI, followed by the same
// deterministic reduce everywhere -- first NaN wins, else the largest value, ties to the lowest
// global index, i.e. np.argmax over the whole set.  RCCL has no MAXLOC, and all-reduce(max) alone
// would lose the index: gather, then reduce.
//
// Two ways to run it, same collective code:
//   * dfh_mgpu_*: one process, N devices -- a context and a host thread per device
//     (ncclCommInitAll); what `python bench.py --gpus N` uses without a launcher;
//   * dfh_comm_*: one process per GPU (ncclGetUniqueId / ncclCommInitRank), the unique id
//     travelling between the processes by whatever the host side has (dragonfly_amd/parallel.py
//     uses a file next to the launcher's rendezvous port).
// RCCL is loaded with dlopen on first use: single-GPU users never pay for (or depend on) it.
#include "common.h"
#include <dlfcn.h>
#include <stdio.h>
#include <unistd.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <mutex>
#include <thread>
#include <rccl/rccl.h>      // types only; every call goes through the table below

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;
bool g_rccl_tried = false;
std::string g_rccl_err;

int rccl_load(const RcclApi** out) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl_tried) {
    g_rccl_tried = true;
    const char* env = getenv("DFH_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) break;
      g_rccl_err = dlerror();
    }
    if (g_rccl.handle) {
      bool ok = true;
      auto sym = [&](const char* s) -> void* {
        void* p = dlsym(g_rccl.handle, s);
        if (!p) { ok = false; g_rccl_err = std::string("missing symbol ") + s; }
        return p;
      };
      g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
      g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
      g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
      g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
      g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
      g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
      g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
      g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
      g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
      g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
      g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
      g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
      if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
    }
  }
  if (!g_rccl.handle) {
    dfh_set_error("RCCL is not available (dlopen librccl.so.1: %s); set DFH_RCCL_LIB", g_rccl_err.c_str());
    return DFH_ERR_HIP;
  }
  *out = &g_rccl;
  return DFH_OK;
}

#define DFH_NCCL(api, call)                                                                  \
  do {                                                                                       \
    ncclResult_t r__ = (call);                                                               \
    if (r__ != ncclSuccess) {                                                                \
      dfh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, (api)->GetErrorString(r__)); \
      return DFH_ERR_HIP;                                                                    \
    }                                                                                        \
  } while (0)

bool pair_better(double va, int64_t ia, double vb, int64_t ib) {
  const bool na = va != va, nb = vb != vb;
  if (na || nb) { if (na && nb) return ia < ib; return na; }
  if (va > vb) return true;
  if (va < vb) return false;
  return ia < ib;
}

struct Pair { double val; int64_t idx; };     // what travels: 16 bytes per rank, bit for bit

// RCCL 2.27 prints a version banner on the process's stdout (C stdio) when a communicator is
// created.  A library must not write to its host's stdout (bench.py's contract: ONE JSON line):
// while a communicator is being initialised, file descriptor 1 points at stderr.
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0) dup2(2, 1);
  }
  ~StdoutToStderr() {
    fflush(stdout);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// host-only pieces of the sharding contract (usable without a GPU; tests/test_mgpu_host.py)
// ---------------------------------------------------------------------------------------------
extern "C" int dfh_shard_bounds(int64_t m, int rank, int world, int64_t align, int64_t* lo, int64_t* hi) {
  DFH_ARG(m >= 0 && world >= 1 && rank >= 0 && rank < world && align >= 1 && lo && hi);
  const int64_t nblk = (m + align - 1) / align;
  const int64_t per = (nblk + world - 1) / world;
  const int64_t a = (int64_t)rank * per * align, b = (int64_t)(rank + 1) * per * align;
  *lo = a < m ? a : m;
  *hi = b < m ? b : m;
  return DFH_OK;
}

extern "C" int dfh_reduce_argmax(const double* vals, const int64_t* idxs, int count, double* best_val,
                                 int64_t* best_idx) {
  DFH_ARG(count >= 0 && (count == 0 || (vals && idxs)) && best_val && best_idx);
  double bv = NAN; int64_t bi = -1;
  for (int r = 0; r < count; ++r) {
    if (idxs[r] < 0) continue;                 // empty shard
    if (bi < 0 || pair_better(vals[r], idxs[r], bv, bi)) { bv = vals[r]; bi = idxs[r]; }
  }
  *best_val = bv; *best_idx = bi;
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------------
// one rank of a communicator, bound to a context (its device, its stream)
// ---------------------------------------------------------------------------------------------
struct dfh_comm {
  dfh_ctx* ctx = nullptr;
  const RcclApi* api = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  bool owns_comm = true;
  Pair* d_send = nullptr;      // device: this rank's pair
  Pair* d_recv = nullptr;      // device: [nranks]
  Pair* h_recv = nullptr;      // pinned: [nranks]
  double* d_red = nullptr;     // device: all-reduce buffer [64]
  double* d_gather = nullptr;  // device: generic small all-gather, (1 + nranks) * GATHER_MAX doubles
};
constexpr int GATHER_MAX = 4096;

static int comm_buffers(dfh_comm* c) {
  DFH_HIP(hipSetDevice(c->ctx->device));
  DFH_HIP(hipMalloc((void**)&c->d_send, sizeof(Pair)));
  DFH_HIP(hipMalloc((void**)&c->d_recv, sizeof(Pair) * c->nranks));
  DFH_HIP(hipHostMalloc((void**)&c->h_recv, sizeof(Pair) * (c->nranks + 1)));
  DFH_HIP(hipMalloc((void**)&c->d_red, sizeof(double) * 64));
  DFH_HIP(hipMalloc((void**)&c->d_gather, sizeof(double) * GATHER_MAX * (size_t)(1 + c->nranks)));
  return DFH_OK;
}

extern "C" int dfh_comm_unique_id(void* id_out) {
  DFH_ARG(id_out != nullptr);
  static_assert(sizeof(ncclUniqueId) == DFH_UNIQUE_ID_BYTES, "unique id size");
  const RcclApi* api = nullptr;
  DFH_TRY(rccl_load(&api));
  ncclUniqueId id;
  DFH_NCCL(api, api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return DFH_OK;
}

extern "C" void dfh_comm_destroy(dfh_comm* c) {
  if (!c) return;
  if (c->ctx && ctx_is_live(c->ctx)) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
  }
  if (c->comm && c->owns_comm && c->api) (void)c->api->CommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->d_gather) (void)hipFree(c->d_gather);
  delete c;
}

extern "C" int dfh_comm_create(dfh_ctx* ctx, int nranks, int rank, const void* id, dfh_comm** out) {
  DFH_ARG(ctx && out && nranks >= 1 && rank >= 0 && rank < nranks && id);
  *out = nullptr;
  const RcclApi* api = nullptr;
  DFH_TRY(rccl_load(&api));
  DFH_HIP(hipSetDevice(ctx->device));
  dfh_comm* c = new dfh_comm();
  c->ctx = ctx; c->api = api; c->rank = rank; c->nranks = nranks;
  auto body = [&]() -> int {
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    StdoutToStderr quiet;
    DFH_NCCL(api, api->CommInitRank(&c->comm, nranks, uid, rank));   // blocks until every rank arrived
    return comm_buffers(c);
  };
  const int rc = body();
  if (rc != DFH_OK) { dfh_comm_destroy(c); return rc; }
  *out = c;
  return DFH_OK;
}

extern "C" int dfh_comm_rank(dfh_comm* c) { return c ? c->rank : -1; }
extern "C" int dfh_comm_size(dfh_comm* c) { return c ? c->nranks : -1; }

// What the communicator ITSELF says (ncclCommCount / ncclCommUserRank / ncclGetVersion), as opposed to what it was
// asked to be: a scaling record can be checked for "RCCL formed N ranks" from these.  A test-mode communicator
// (duplicate devices: the exchange runs on the host) reports 0 ranks formed.
extern "C" int dfh_comm_info(dfh_comm* c, int32_t* ranks_formed, int32_t* rank, int32_t* rccl_version) {
  DFH_ARG(c && ranks_formed && rank && rccl_version);
  *ranks_formed = 0; *rank = c->rank; *rccl_version = 0;
  if (!c->comm || !c->api) return DFH_OK;
  int cnt = 0, ur = -1, ver = 0;
  DFH_NCCL(c->api, c->api->CommCount(c->comm, &cnt));
  DFH_NCCL(c->api, c->api->CommUserRank(c->comm, &ur));
  if (c->api->GetVersion) c->api->GetVersion(&ver);
  *ranks_formed = cnt; *rank = ur; *rccl_version = ver;
  return DFH_OK;
}

// enqueue the all-gather of this rank's pair on its context's stream
static int comm_gather_enqueue(dfh_comm* c, double val, int64_t idx) {
  DFH_HIP(hipSetDevice(c->ctx->device));
  c->h_recv[c->nranks].val = val;
  c->h_recv[c->nranks].idx = idx;
  DFH_HIP(hipMemcpyAsync(c->d_send, &c->h_recv[c->nranks], sizeof(Pair), hipMemcpyHostToDevice, c->ctx->stream));
  DFH_NCCL(c->api, c->api->AllGather(c->d_send, c->d_recv, sizeof(Pair), ncclChar, c->comm, c->ctx->stream));
  DFH_HIP(hipMemcpyAsync(c->h_recv, c->d_recv, sizeof(Pair) * c->nranks, hipMemcpyDeviceToHost, c->ctx->stream));
  return DFH_OK;
}
static int comm_gather_finish(dfh_comm* c, double* best_val, int64_t* best_idx) {
  DFH_HIP(hipSetDevice(c->ctx->device));
  DFH_HIP(hipStreamSynchronize(c->ctx->stream));
  double bv = NAN; int64_t bi = -1;
  for (int r = 0; r < c->nranks; ++r) {
    if (c->h_recv[r].idx < 0) continue;
    if (bi < 0 || pair_better(c->h_recv[r].val, c->h_recv[r].idx, bv, bi)) { bv = c->h_recv[r].val; bi = c->h_recv[r].idx; }
  }
  if (best_val) *best_val = bv;
  if (best_idx) *best_idx = bi;
  return DFH_OK;
}

extern "C" int dfh_comm_allgather_argmax(dfh_comm* c, double local_val, int64_t local_idx, double* best_val,
                                         int64_t* best_idx) {
  DFH_ARG(c && c->comm);
  DFH_TRY(comm_gather_enqueue(c, local_val, local_idx));
  return comm_gather_finish(c, best_val, best_idx);
}

extern "C" int dfh_comm_allreduce_max(dfh_comm* c, double* inout, int count) {
  DFH_ARG(c && c->comm && inout && count >= 1 && count <= 64);
  DFH_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  DFH_HIP(hipMemcpyAsync(c->d_red, inout, sizeof(double) * count, hipMemcpyHostToDevice, s));
  DFH_NCCL(c->api, c->api->AllReduce(c->d_red, c->d_red, (size_t)count, ncclDouble, ncclMax, c->comm, s));
  DFH_HIP(hipMemcpyAsync(inout, c->d_red, sizeof(double) * count, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

// recv[r * count + j] = rank r's send[j]  (host buffers; e.g. the winning candidate's coordinates
// travelling from its owner)
extern "C" int dfh_comm_allgather_f64(dfh_comm* c, const double* send, int count, double* recv) {
  DFH_ARG(c && c->comm && send && recv && count >= 1 && count <= GATHER_MAX);
  DFH_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  double* d_send = c->d_gather;
  double* d_recv = c->d_gather + GATHER_MAX;
  DFH_HIP(hipMemcpyAsync(d_send, send, sizeof(double) * count, hipMemcpyHostToDevice, s));
  DFH_NCCL(c->api, c->api->AllGather(d_send, d_recv, (size_t)count, ncclDouble, c->comm, s));
  DFH_HIP(hipMemcpyAsync(recv, d_recv, sizeof(double) * count * c->nranks, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

extern "C" int dfh_comm_barrier(dfh_comm* c) {
  DFH_ARG(c && c->comm);
  DFH_TRY(dfh_sync(c->ctx));
  double z = 0.0;
  return dfh_comm_allreduce_max(c, &z, 1);
}

// ---------------------------------------------------------------------------------------------
// one process, N devices
// ---------------------------------------------------------------------------------------------
struct dfh_mgpu {
  int n = 0;
  bool host_exchange = false;   // test mode (DFH_MGPU_ALLOW_DUPLICATE_DEVICES): no communicator, pairs reduced on the host
  std::vector<int> devices;
  std::vector<dfh_ctx*> ctxs;
  std::vector<dfh_comm*> comms;
  std::vector<dfh_gp*> gps;
};

extern "C" void dfh_mgpu_destroy(dfh_mgpu* mg) {
  if (!mg) return;
  for (auto g : mg->gps) if (g) dfh_gp_free(g);
  for (auto c : mg->comms) dfh_comm_destroy(c);
  for (auto c : mg->ctxs) dfh_ctx_destroy(c);
  delete mg;
}

extern "C" int dfh_mgpu_create(int n_devices, const int* device_ids, dfh_mgpu** out) {
  DFH_ARG(out && n_devices >= 1 && n_devices <= DFH_MAX_DEVICES);
  *out = nullptr;
  int visible = 0;
  DFH_HIP(hipGetDeviceCount(&visible));
  // Test switch (tests/test_gpu_mgpu.py): with DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 a device id may be
  // given several times -- N contexts and N host threads on ONE device, driven concurrently through
  // the same fan-out as N devices.  RCCL refuses duplicate devices in a communicator, so the 16-byte
  // pairs are then reduced on the host (the same reduce every rank applies to the gathered pairs).
  const char* dup_env = getenv("DFH_MGPU_ALLOW_DUPLICATE_DEVICES");
  bool duplicates = false;
  if (dup_env && atoi(dup_env) != 0 && device_ids)
    for (int r = 0; r < n_devices; ++r)
      for (int q = 0; q < r; ++q) duplicates = duplicates || device_ids[q] == device_ids[r];
  if (!duplicates && n_devices > visible) {
    dfh_set_error("dfh_mgpu_create: %d devices requested, %d visible", n_devices, visible);
    return DFH_ERR_BAD_ARG;
  }
  const RcclApi* api = nullptr;
  if (!duplicates) {
    // one device has nobody to exchange with: RCCL is used when it is there (the same code path as
    // N devices), but its absence is not an error
    const int rc = rccl_load(&api);
    if (rc != DFH_OK && n_devices > 1) return rc;
    if (rc != DFH_OK) api = nullptr;
  }
  dfh_mgpu* mg = new dfh_mgpu();
  mg->n = n_devices;
  mg->host_exchange = duplicates;
  auto body = [&]() -> int {
    for (int r = 0; r < n_devices; ++r) {
      const int dev = device_ids ? device_ids[r] : r;
      DFH_ARG(dev >= 0 && dev < visible);
      if (!duplicates) for (int q = 0; q < r; ++q) DFH_ARG(mg->devices[q] != dev);     // one rank per device
      mg->devices.push_back(dev);
      dfh_ctx* ctx = nullptr;
      DFH_TRY(dfh_ctx_create(dev, &ctx));
      mg->ctxs.push_back(ctx);
    }
    mg->gps.assign((size_t)n_devices, nullptr);
    if (!api) return DFH_OK;                  // single device without RCCL: no communicator
    std::vector<ncclComm_t> cs((size_t)n_devices, nullptr);
    {
      StdoutToStderr quiet;
      DFH_NCCL(api, api->CommInitAll(cs.data(), n_devices, mg->devices.data()));
    }
    for (int r = 0; r < n_devices; ++r) {
      dfh_comm* c = new dfh_comm();
      c->ctx = mg->ctxs[r]; c->api = api; c->comm = cs[r]; c->rank = r; c->nranks = n_devices;
      mg->comms.push_back(c);
      DFH_TRY(comm_buffers(c));
    }
    mg->gps.assign((size_t)n_devices, nullptr);
    return DFH_OK;
  };
  const int rc = body();
  if (rc != DFH_OK) { dfh_mgpu_destroy(mg); return rc; }
  *out = mg;
  return DFH_OK;
}

extern "C" int dfh_mgpu_size(dfh_mgpu* mg) { return mg ? mg->n : -1; }
extern "C" dfh_ctx* dfh_mgpu_ctx(dfh_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < mg->n) ? mg->ctxs[rank] : nullptr;
}
extern "C" dfh_gp* dfh_mgpu_gp(dfh_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < mg->n) ? mg->gps[rank] : nullptr;
}
extern "C" dfh_comm* dfh_mgpu_comm(dfh_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < (int)mg->comms.size()) ? mg->comms[rank] : nullptr;
}

namespace {

// run fn(rank) on one host thread per device (the calls below block on their device's streams);
// the first failure's status and message are handed to the caller's thread
template <typename Fn>
int fan_out(dfh_mgpu* mg, Fn fn) {
  const int n = mg->n;
  std::vector<int> rcs((size_t)n, DFH_OK);
  std::vector<std::string> errs((size_t)n);
  auto work = [&](int r) {
    if (hipSetDevice(mg->devices[r]) != hipSuccess) { rcs[r] = DFH_ERR_HIP; errs[r] = "hipSetDevice failed"; return; }
    rcs[r] = fn(r);
    if (rcs[r] != DFH_OK) errs[r] = dfh_last_error();
  };
  if (n == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 1; r < n; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < n; ++r)
    if (rcs[r] != DFH_OK) { dfh_set_error("device %d: %s", mg->devices[r], errs[r].c_str()); return rcs[r]; }
  return DFH_OK;
}

// the exchange: every device enqueues its all-gather inside one RCCL group (single-thread,
// multi-device form), then every rank reduces its own copy; rank 0's answer is returned after
// checking that the others hold the same
int exchange(dfh_mgpu* mg, const std::vector<double>& vals, const std::vector<int64_t>& idxs, double* best_val,
             int64_t* best_idx) {
  if (mg->comms.empty()) {                    // single device without RCCL, or the duplicate-device test mode: the reduce alone
    DFH_ARG(mg->n == 1 || mg->host_exchange);
    return dfh_reduce_argmax(vals.data(), idxs.data(), mg->n, best_val, best_idx);
  }
  const RcclApi* api = mg->comms[0]->api;
  DFH_NCCL(api, api->GroupStart());
  int rc = DFH_OK;
  for (int r = 0; r < mg->n && rc == DFH_OK; ++r) rc = comm_gather_enqueue(mg->comms[r], vals[r], idxs[r]);
  DFH_NCCL(api, api->GroupEnd());
  DFH_TRY(rc);
  double bv0 = NAN; int64_t bi0 = -1;
  for (int r = 0; r < mg->n; ++r) {
    double bv; int64_t bi;
    DFH_TRY(comm_gather_finish(mg->comms[r], &bv, &bi));
    if (r == 0) { bv0 = bv; bi0 = bi; continue; }
    if (bi != bi0 || memcmp(&bv, &bv0, sizeof(double)) != 0) {
      dfh_set_error("multi-GPU arg-max: rank %d reduced to (%.17g, %lld), rank 0 to (%.17g, %lld)", r, bv,
                    (long long)bi, bv0, (long long)bi0);
      return DFH_ERR_HIP;
    }
  }
  if (best_val) *best_val = bv0;
  if (best_idx) *best_idx = bi0;
  return DFH_OK;
}

}  // namespace

// Replicated fit: dfh_gp_fit on every device (GP.build_posterior, gp_core.py:155-163).  X[r] /
// y_centred[r] are rank r's copies (host pointers, or device pointers on rank r's device); the
// same host pointer may be given for every rank.  lml / jitter_power: [n_devices] or NULL.
extern "C" int dfh_mgpu_fit(dfh_mgpu* mg, const dfh_kernel_desc* k, const double* const* X, int64_t n, int64_t d,
                            const double* const* y_centred, double noise_var, int flags, double* lml,
                            int32_t* jitter_power) {
  DFH_ARG(mg && k && X && y_centred);
  for (int r = 0; r < mg->n; ++r) DFH_ARG(X[r] && y_centred[r]);
  return fan_out(mg, [&](int r) -> int {
    if (mg->gps[r]) { dfh_gp_free(mg->gps[r]); mg->gps[r] = nullptr; }
    double l = 0.0; int32_t jp = INT32_MIN;
    const int rc = dfh_gp_fit(mg->ctxs[r], k, X[r], n, d, y_centred[r], noise_var, flags, &mg->gps[r], &l, &jp);
    if (lml) lml[r] = l;
    if (jitter_power) jitter_power[r] = jp;
    return rc;
  });
}

extern "C" int dfh_mgpu_free_fit(dfh_mgpu* mg) {
  DFH_ARG(mg != nullptr);
  for (auto& g : mg->gps) { if (g) dfh_gp_free(g); g = nullptr; }
  return DFH_OK;
}

// Blocked-joint Thompson sampling (dfh_gp_ts) over contiguous candidate shards: rank r holds rows
// [off_r, off_r + m[r]) of the global candidate set, off_r = m[0] + ... + m[r-1] (cut the set on
// multiples of `block` -- dfh_shard_bounds with align = block -- and the blocks, hence the draw,
// are those of one device doing it all).  U[r]: the shard's standard normals.  m[r] == 0 is an
// empty shard.  best_idx is the GLOBAL row index.  local_vals / local_idx: optional [n_devices].
extern "C" int dfh_mgpu_ts(dfh_mgpu* mg, const double* const* Xs, const int64_t* m, int64_t block,
                           const double* const* U, double mean_const, double* best_val, int64_t* best_idx,
                           double* local_vals, int64_t* local_idx) {
  DFH_ARG(mg && Xs && m && U && block >= 1);
  std::vector<int64_t> off((size_t)mg->n + 1, 0);
  for (int r = 0; r < mg->n; ++r) {
    DFH_ARG(m[r] >= 0 && (m[r] == 0 || (Xs[r] && U[r])) && mg->gps[r]);
    off[r + 1] = off[r] + m[r];
  }
  std::vector<double> vals((size_t)mg->n, NAN);
  std::vector<int64_t> idxs((size_t)mg->n, -1);
  DFH_TRY(fan_out(mg, [&](int r) -> int {
    if (m[r] == 0) return DFH_OK;
    double v = NAN; int64_t i = -1;
    DFH_TRY(dfh_gp_ts(mg->gps[r], Xs[r], m[r], block, U[r], mean_const, nullptr, nullptr, &v, &i, nullptr));
    vals[r] = v; idxs[r] = off[r] + i;
    return DFH_OK;
  }));
  for (int r = 0; r < mg->n; ++r) {
    if (local_vals) local_vals[r] = vals[r];
    if (local_idx) local_idx[r] = idxs[r];
  }
  return exchange(mg, vals, idxs, best_val, best_idx);
}

// Fused posterior + acquisition + arg-max (dfh_gp_acq_argmax) over contiguous candidate shards.
extern "C" int dfh_mgpu_acq_argmax(dfh_mgpu* mg, int acq, const double* params, const double* const* Xs,
                                   const int64_t* m, double mean_const, double* best_val, int64_t* best_idx,
                                   double* local_vals, int64_t* local_idx) {
  DFH_ARG(mg && Xs && m);
  std::vector<int64_t> off((size_t)mg->n + 1, 0);
  for (int r = 0; r < mg->n; ++r) {
    DFH_ARG(m[r] >= 0 && (m[r] == 0 || Xs[r]) && mg->gps[r]);
    off[r + 1] = off[r] + m[r];
  }
  std::vector<double> vals((size_t)mg->n, NAN);
  std::vector<int64_t> idxs((size_t)mg->n, -1);
  DFH_TRY(fan_out(mg, [&](int r) -> int {
    if (m[r] == 0) return DFH_OK;
    double v = NAN; int64_t i = -1;
    DFH_TRY(dfh_gp_acq_argmax(mg->gps[r], acq, params, Xs[r], m[r], nullptr, 0, mean_const, nullptr, nullptr, &v, &i));
    vals[r] = v; idxs[r] = off[r] + i;
    return DFH_OK;
  }));
  for (int r = 0; r < mg->n; ++r) {
    if (local_vals) local_vals[r] = vals[r];
    if (local_idx) local_idx[r] = idxs[r];
  }
  return exchange(mg, vals, idxs, best_val, best_idx);
}

// The exchange on its own: per-rank (value, global index) pairs in, the reduced pair out.
extern "C" int dfh_mgpu_allgather_argmax(dfh_mgpu* mg, const double* vals, const int64_t* idxs, double* best_val,
                                         int64_t* best_idx) {
  DFH_ARG(mg && vals && idxs);
  std::vector<double> v(vals, vals + mg->n);
  std::vector<int64_t> i(idxs, idxs + mg->n);
  return exchange(mg, v, i, best_val, best_idx);
}

extern "C" int dfh_mgpu_sync(dfh_mgpu* mg) {
  DFH_ARG(mg != nullptr);
  for (int r = 0; r < mg->n; ++r) {
    DFH_HIP(hipSetDevice(mg->devices[r]));
    DFH_TRY(dfh_sync(mg->ctxs[r]));
  }
  return DFH_OK;
}
