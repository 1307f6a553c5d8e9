// rccl_exchange.cpp -- the multi-GPU exchanges of the path, native: RCCL over xGMI from inside libmpfitch.so.
//
// north_star: "bootstrap replicates and independent SPR start-trees shard embarrassingly across the 8 GPUs of one node with a
// single RCCL all-reduce of best scores per round".  Two collectives are all the path has:
//   * mpf_rccl_exchange -- an mpf_ufb_exchange_fn (include/mpfitch.h): the (candidate, sample, score) events of one scan batch
//     of a sample-sharded online UFBoot phase (every rank books its own samples' REPS; every rank replays ALL events, so the one
//     search chain stays identical everywhere).  One all-gather of fixed-size blocks [count, tag | kEventBlock triples] per rank
//     on a stream of its own; only a batch in which some rank has more events pays a second, exactly sized one.  Device staging
//     buffers, pinned host mirrors, no Python and no torch in between (mpboot_amd/shard.py's gather_events is the same protocol
//     over torch.distributed -- gloo in the CPU tests);
//   * mpf_rccl_allreduce_min -- best scores (and, packed beside them, their owners) of independent units.
// librccl is opened at run time (dlopen): libmpfitch.so loads, and every single-GPU path works, on a machine without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mpfitch.h"

namespace mpf {
void set_error(const std::string &msg);
}

namespace {

constexpr uint32_t kEventBlock = 4096;            // (mpboot_amd/shard.py: EVENT_BLOCK)
constexpr size_t kBlockWords = (size_t)(kEventBlock + 1) * 3;

typedef struct { char internal[128]; } rcclUniqueId;     // ncclUniqueId (rccl.h:40-43)
typedef void *rcclComm;
struct Api {
  void *lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId *) = nullptr;
  int (*CommInitRank)(rcclComm *, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, rcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int kNcclUint32 = 3, kNcclMax = 2, kNcclMin = 3;      // ncclDataType_t / ncclRedOp_t (rccl.h:451, :462)

Api &api()
{
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    // The RCCL that belongs to the HIP runtime THIS library runs on: a process that has imported torch holds a second ROCm stack
    // (torch/lib/libamdhip64.so + librccl.so, other sonames), and a plain dlopen("librccl.so") would hand back that one -- whose
    // runtime knows nothing of this library's device buffers and streams.  So: the librccl next to the libamdhip64 our own HIP
    // calls resolve to, by path; the plain names only where that fails.
    std::vector<std::string> names;
    Dl_info di;
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t cut = dir.find_last_of('/');
      if (cut != std::string::npos) {
        dir.resize(cut);
        names.push_back(dir + "/librccl.so.1");
        names.push_back(dir + "/librccl.so");
      }
    }
    for (const char *name : {"/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"}) names.push_back(name);
    for (const std::string &name : names) {
      a.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (!a.lib) return;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce;
  });
  return a;
}

int fail(const char *what, int rc)
{
  const Api &a = api();
  mpf::set_error(std::string("RCCL: ") + what + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "error " + std::to_string(rc)));
  return MPF_E_HIP;
}

}  // namespace

struct mpf_rccl {
  rcclComm comm = nullptr;
  int rank = 0, world = 1, dev = 0;
  hipStream_t st = nullptr;
  uint32_t *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;      // one block / world blocks
  uint32_t *d_over_s = nullptr, *d_over_r = nullptr, *h_over_r = nullptr;                   // overflow, grown on demand
  size_t over_cap = 0;                              // triples per rank
  std::vector<mpf_ufb_event> merged;
  uint64_t n_exchanges = 0, n_overflows = 0;
};

#define RHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) { mpf::set_error(std::string(#expr) + ": " + hipGetErrorString(e__)); return MPF_E_HIP; } \
  } while (0)

extern "C" {

int mpf_rccl_available(void) { return api().ok ? 1 : 0; }

int mpf_rccl_unique_id(uint8_t *out /* [128] */)
{
  if (!out) { mpf::set_error("mpf_rccl_unique_id: bad argument"); return MPF_E_INVALID; }
  if (!api().ok) { mpf::set_error("librccl.so is not available"); return MPF_E_UNSUPPORTED; }
  rcclUniqueId id;
  const int rc = api().GetUniqueId(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  std::memcpy(out, id.internal, sizeof(id.internal));
  return MPF_OK;
}

void mpf_rccl_destroy(mpf_rccl *c)
{
  if (!c) return;
  (void)hipSetDevice(c->dev);
  if (c->st) (void)hipStreamSynchronize(c->st);
  if (c->comm && api().ok) (void)api().CommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  if (c->d_over_s) (void)hipFree(c->d_over_s);
  if (c->d_over_r) (void)hipFree(c->d_over_r);
  if (c->h_over_r) (void)hipHostFree(c->h_over_r);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

// one communicator per process (= per GPU): `id` from rank 0's mpf_rccl_unique_id, carried to the others by whatever the host
// has (a file, MPI, torch.distributed's store)
int mpf_rccl_create(mpf_rccl **out, const uint8_t *id /* [128] */, int32_t rank, int32_t world, int32_t device)
{
  if (!out || !id || world < 1 || rank < 0 || rank >= world) { mpf::set_error("mpf_rccl_create: bad argument"); return MPF_E_INVALID; }
  if (!api().ok) { mpf::set_error("librccl.so is not available"); return MPF_E_UNSUPPORTED; }
  RHIP(hipSetDevice(device));
  mpf_rccl *c = new mpf_rccl();
  c->rank = rank; c->world = world; c->dev = device;
  rcclUniqueId uid;
  std::memcpy(uid.internal, id, sizeof(uid.internal));
  int rc = api().CommInitRank(&c->comm, world, uid, rank);
  if (rc) { mpf_rccl_destroy(c); return fail("ncclCommInitRank", rc); }
  hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_send, kBlockWords * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_recv, kBlockWords * 4 * (size_t)world);
  if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_send, kBlockWords * 4, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_recv, kBlockWords * 4 * (size_t)world, hipHostMallocDefault);
  if (e != hipSuccess) { mpf::set_error(std::string("mpf_rccl_create: ") + hipGetErrorString(e)); mpf_rccl_destroy(c); return MPF_E_HIP; }
  *out = c;
  return MPF_OK;
}

// mpf_ufb_exchange_fn: arg = the mpf_rccl of this rank.  Ranks whose tags differ are not at the same point of the run: non-zero.
int mpf_rccl_exchange(void *arg, uint32_t tag, const mpf_ufb_event *local, uint32_t n_local, const mpf_ufb_event **all, uint32_t *n_all)
{
  mpf_rccl *c = static_cast<mpf_rccl *>(arg);
  if (!c || !all || !n_all) return 1;
  (void)hipSetDevice(c->dev);
  c->n_exchanges++;
  const uint32_t head = n_local < kEventBlock ? n_local : kEventBlock;
  c->h_send[0] = n_local; c->h_send[1] = tag; c->h_send[2] = 0;
  if (head) std::memcpy(c->h_send + 3, local, (size_t)head * sizeof(mpf_ufb_event));
  if (hipMemcpyAsync(c->d_send, c->h_send, (3 + (size_t)head * 3) * 4, hipMemcpyHostToDevice, c->st) != hipSuccess) return 1;
  if (api().AllGather(c->d_send, c->d_recv, kBlockWords, kNcclUint32, c->comm, c->st)) return 1;
  // (the headers first would save nothing: the blocks are 48 KB, one copy brings everything)
  if (hipMemcpyAsync(c->h_recv, c->d_recv, kBlockWords * 4 * (size_t)c->world, hipMemcpyDeviceToHost, c->st) != hipSuccess) return 1;
  if (hipStreamSynchronize(c->st) != hipSuccess) return 1;
  uint32_t over = 0;
  size_t total = 0;
  for (int r = 0; r < c->world; r++) {
    const uint32_t *b = c->h_recv + (size_t)r * kBlockWords;
    if (b[1] != tag) return 2;                     // out of step
    total += b[0];
    if (b[0] > kEventBlock) over = std::max(over, b[0] - kEventBlock);
  }
  if (over) {
    // (rare: more events than a block holds -- the remainder in an all-gather of exactly the size the largest rank needs)
    c->n_overflows++;
    if (over > c->over_cap) {
      // Every rank sees the same `over` and keeps the same over_cap, so all of them come here together.  An allocation that fails
      // on ONE rank must not leave the others waiting in the gather below: the ranks agree on the outcome first (one all-reduce of
      // a status word; this path runs once per growth of the buffers) and return non-zero together.
      if (c->d_over_s) (void)hipFree(c->d_over_s);
      if (c->d_over_r) (void)hipFree(c->d_over_r);
      if (c->h_over_r) (void)hipHostFree(c->h_over_r);
      c->d_over_s = c->d_over_r = c->h_over_r = nullptr;
      c->over_cap = 0;
      const size_t cap = (size_t)over + over / 2 + 64;
      uint32_t bad = 0;
      if (hipMalloc((void **)&c->d_over_s, cap * 12) != hipSuccess) bad = 1;
      if (!bad && hipMalloc((void **)&c->d_over_r, cap * 12 * (size_t)c->world) != hipSuccess) bad = 1;
      if (!bad && hipHostMalloc((void **)&c->h_over_r, cap * 12 * (size_t)c->world, hipHostMallocDefault) != hipSuccess) bad = 1;
      c->h_send[0] = bad;
      if (hipMemcpyAsync(c->d_send, c->h_send, 4, hipMemcpyHostToDevice, c->st) != hipSuccess) return 1;
      if (api().AllReduce(c->d_send, c->d_send, 1, kNcclUint32, kNcclMax, c->comm, c->st)) return 1;
      if (hipMemcpyAsync(c->h_send, c->d_send, 4, hipMemcpyDeviceToHost, c->st) != hipSuccess) return 1;
      if (hipStreamSynchronize(c->st) != hipSuccess) return 1;
      if (c->h_send[0]) {
        if (c->d_over_s) (void)hipFree(c->d_over_s);
        if (c->d_over_r) (void)hipFree(c->d_over_r);
        if (c->h_over_r) (void)hipHostFree(c->h_over_r);
        c->d_over_s = c->d_over_r = c->h_over_r = nullptr;
        return 3;                                  // some rank is out of memory: every rank leaves here
      }
      c->over_cap = cap;
    }
    if (n_local > kEventBlock &&
        hipMemcpyAsync(c->d_over_s, local + kEventBlock, (size_t)(n_local - kEventBlock) * 12, hipMemcpyHostToDevice, c->st) != hipSuccess) return 1;
    if (api().AllGather(c->d_over_s, c->d_over_r, (size_t)over * 3, kNcclUint32, c->comm, c->st)) return 1;
    if (hipMemcpyAsync(c->h_over_r, c->d_over_r, (size_t)over * 12 * (size_t)c->world, hipMemcpyDeviceToHost, c->st) != hipSuccess) return 1;
    if (hipStreamSynchronize(c->st) != hipSuccess) return 1;
  }
  c->merged.clear();
  c->merged.reserve(total);
  for (int r = 0; r < c->world; r++) {
    const uint32_t *b = c->h_recv + (size_t)r * kBlockWords;
    const uint32_t n = b[0], nh = n < kEventBlock ? n : kEventBlock;
    const mpf_ufb_event *ev = reinterpret_cast<const mpf_ufb_event *>(b + 3);
    c->merged.insert(c->merged.end(), ev, ev + nh);
    if (n > kEventBlock) {
      const mpf_ufb_event *eo = reinterpret_cast<const mpf_ufb_event *>(c->h_over_r + (size_t)r * (size_t)over * 3);
      c->merged.insert(c->merged.end(), eo, eo + (n - kEventBlock));
    }
  }
  *all = c->merged.empty() ? nullptr : c->merged.data();
  *n_all = (uint32_t)c->merged.size();
  return 0;
}

// element-wise minimum over the ranks (best scores of independent units; pack the owner beside the score -- score << 8 | rank --
// to learn who holds it): the "single all-reduce of best scores per round"
int mpf_rccl_allreduce_min(mpf_rccl *c, uint32_t *vals, int32_t n)
{
  if (!c || !vals || n < 1) { mpf::set_error("mpf_rccl_allreduce_min: bad argument"); return MPF_E_INVALID; }
  RHIP(hipSetDevice(c->dev));
  uint32_t *d = nullptr;
  RHIP(hipMalloc((void **)&d, (size_t)n * 4));
  hipError_t e = hipMemcpyAsync(d, vals, (size_t)n * 4, hipMemcpyHostToDevice, c->st);
  int rc = 0;
  if (e == hipSuccess) rc = api().AllReduce(d, d, (size_t)n, kNcclUint32, kNcclMin, c->comm, c->st);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(vals, d, (size_t)n * 4, hipMemcpyDeviceToHost, c->st);
  if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->st);
  (void)hipFree(d);
  if (rc) return fail("ncclAllReduce", rc);
  if (e != hipSuccess) { mpf::set_error(std::string("mpf_rccl_allreduce_min: ") + hipGetErrorString(e)); return MPF_E_HIP; }
  return MPF_OK;
}

int mpf_rccl_counters(const mpf_rccl *c, uint64_t *exchanges, uint64_t *overflows)
{
  if (!c) return MPF_E_INVALID;
  if (exchanges) *exchanges = c->n_exchanges;
  if (overflows) *overflows = c->n_overflows;
  return MPF_OK;
}

}  // extern "C"
