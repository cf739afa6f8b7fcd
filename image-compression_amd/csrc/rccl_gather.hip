// rccl_gather.hip -- the one collective of the multi-GPU path: the gather of the compressed output to one rank over RCCL
// (include/ic_amd.h "multi-GPU, ONE PROCESS PER GPU").  Host code only: it binds librccl at run time and enqueues
// one grouped send / receive exchange on the caller's stream.
//
// Why dlopen and not a link dependency: the encoders need no collective at all, a process that never gathers must not need
// RCCL (573 MB) to load libic_amd.so, and a process that already carries an RCCL -- a PyTorch process carries its own copy --
// must use THAT copy: two RCCLs in one process would each bring their own proxy threads and IPC set-up.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the pointers bound below

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "ic_abi.h"

namespace {
using icamd::fail;
using icamd::kErrorChars;

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  char why[kErrorChars] = "";  // why binding failed
  bool ok = false;
};

// Bound once per process; never unloaded (RCCL's own threads may outlive any point at which we could).
Rccl &rccl() {
  static Rccl *r = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    r = new Rccl();
    // 1. a copy that is already in the process (PyTorch's librccl.so has the same soname); 2. ICAMD_RCCL_LIBRARY (a path);
    // 3. the loader's search path (ROCm's lib directory is on libic_amd.so's rpath)
    const char *names[4] = { nullptr, nullptr, nullptr, nullptr };
    int n = 0;
    const char *env = getenv("ICAMD_RCCL_LIBRARY");
    r->handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!r->handle) r->handle = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!r->handle) {
      if (env && *env) names[n++] = env;
      names[n++] = "librccl.so.1";
      names[n++] = "librccl.so";
      for (int i = 0; i < n && !r->handle; ++i) r->handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (!r->handle) {
      const char *e = dlerror();
      std::snprintf(r->why, sizeof r->why, "librccl could not be loaded: %s", e ? e : "dlopen failed");
      return;
    }
    bool all = true;
    auto bind = [&](auto &slot, const char *name) {
      slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(dlsym(r->handle, name));
      if (!slot) {
        all = false;
        std::snprintf(r->why, sizeof r->why, "librccl lacks %s", name);
      }
    };
    bind(r->GetUniqueId, "ncclGetUniqueId");
    bind(r->CommInitRank, "ncclCommInitRank");
    bind(r->CommDestroy, "ncclCommDestroy");
    bind(r->GroupStart, "ncclGroupStart");
    bind(r->GroupEnd, "ncclGroupEnd");
    bind(r->Send, "ncclSend");
    bind(r->Recv, "ncclRecv");
    bind(r->GetErrorString, "ncclGetErrorString");
    r->ok = all;
  });
  return *r;
}

int need_rccl(Rccl **out) {
  Rccl &r = rccl();
  if (!r.ok) return fail(ICAMD_ERR_NO_DEVICE, r.why);
  *out = &r;
  return ICAMD_OK;
}

int fail_rccl(const Rccl &r, const char *what, ncclResult_t res) {
  char buf[kErrorChars];
  std::snprintf(buf, sizeof buf, "%s: %s", what, r.GetErrorString ? r.GetErrorString(res) : "RCCL error");
  return fail(ICAMD_ERR_HIP, buf);
}

static_assert(ICAMD_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id buffer of include/ic_amd.h is an ncclUniqueId");
static_assert(sizeof(ncclUniqueId) == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId is its 128 opaque bytes");

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int icamd_rccl_available(void) try {
  Rccl *r = nullptr;
  return need_rccl(&r) == ICAMD_OK ? 1 : 0;
} catch (...) {
  (void)icamd::abi_exception();
  return 0;
}

int icamd_rccl_get_unique_id(void *id) try {
  if (!id) return fail(ICAMD_ERR_ARG, "icamd_rccl_get_unique_id: null buffer");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc != ICAMD_OK) return rc;
  ncclUniqueId uid;
  const ncclResult_t res = r->GetUniqueId(&uid);
  if (res != ncclSuccess) return fail_rccl(*r, "ncclGetUniqueId", res);
  std::memcpy(id, &uid, sizeof uid);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_rccl_comm_init(void **comm, int world, int rank, const void *id) try {
  if (!comm || !id) return fail(ICAMD_ERR_ARG, "icamd_rccl_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(ICAMD_ERR_ARG, "icamd_rccl_comm_init: rank outside the world");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc != ICAMD_OK) return rc;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ICAMD_ERR_NO_DEVICE, "no HIP device available");
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclComm_t c = nullptr;
  const ncclResult_t res = r->CommInitRank(&c, world, uid, rank);
  if (res != ncclSuccess) return fail_rccl(*r, "ncclCommInitRank", res);
  *comm = c;
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_rccl_comm_destroy(void *comm) try {
  if (!comm) return ICAMD_OK;
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc != ICAMD_OK) return rc;
  const ncclResult_t res = r->CommDestroy(static_cast<ncclComm_t>(comm));
  if (res != ncclSuccess) return fail_rccl(*r, "ncclCommDestroy", res);
  return ICAMD_OK;
} ICAMD_ABI_CATCH

int icamd_gather_blocks_rccl(void *comm, int rank, int world, int root, const size_t *counts_bytes, const void *d_local,
                             void *d_root_buffer, const size_t *root_offsets_bytes, void *hip_stream) try {
  if (!comm || !counts_bytes) return fail(ICAMD_ERR_ARG, "icamd_gather_blocks_rccl: null communicator or counts");
  if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
    return fail(ICAMD_ERR_ARG, "icamd_gather_blocks_rccl: rank / root outside the world");
  if (counts_bytes[rank] != 0 && !d_local) return fail(ICAMD_ERR_ARG, "icamd_gather_blocks_rccl: bytes to contribute but no d_local");
  size_t total = 0;
  for (int p = 0; p < world; ++p) total += counts_bytes[p];
  if (rank == root && total != 0 && !d_root_buffer) return fail(ICAMD_ERR_ARG, "icamd_gather_blocks_rccl: root without a buffer");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc != ICAMD_OK) return rc;
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  // (ICAMD_RCCL_SELF_SENDRECV=1: test knob -- root's own range travels through ncclSend / ncclRecv to itself instead of the
  // device-to-device copy, so that a world of ONE rank on a 1-GPU box executes RCCL's send / receive path, not only its set-up)
  static const bool self_sendrecv = [] { const char *e = getenv("ICAMD_RCCL_SELF_SENDRECV"); return e && e[0] == '1'; }();

  if (rank != root) {
    if (counts_bytes[rank] == 0) return ICAMD_OK;  // nothing of this rank's is expected on root (root skips it too)
    const ncclResult_t res = r->Send(d_local, counts_bytes[rank], ncclUint8, root, c, stream);
    if (res != ncclSuccess) return fail_rccl(*r, "ncclSend", res);
    return ICAMD_OK;
  }
  // root: one group of receives, every peer writes its own range of this GPU's HBM
  uint8_t *base = static_cast<uint8_t *>(d_root_buffer);
  auto slot_of = [&](int p) {
    if (root_offsets_bytes) return base + root_offsets_bytes[p];
    size_t off = 0;
    for (int q = 0; q < p; ++q) off += counts_bytes[q];
    return base + off;
  };
  ncclResult_t res = r->GroupStart();
  if (res != ncclSuccess) return fail_rccl(*r, "ncclGroupStart", res);
  ncclResult_t first_bad = ncclSuccess;
  const char *bad_what = "";
  for (int p = 0; p < world; ++p) {
    if (counts_bytes[p] == 0 || (p == root && !self_sendrecv)) continue;
    res = r->Recv(slot_of(p), counts_bytes[p], ncclUint8, p, c, stream);
    if (res != ncclSuccess && first_bad == ncclSuccess) { first_bad = res; bad_what = "ncclRecv"; }
  }
  if (self_sendrecv && counts_bytes[root] != 0) {
    res = r->Send(d_local, counts_bytes[root], ncclUint8, root, c, stream);
    if (res != ncclSuccess && first_bad == ncclSuccess) { first_bad = res; bad_what = "ncclSend (self)"; }
  }
  res = r->GroupEnd();  // (always closed, also after a failing call inside the group)
  if (first_bad != ncclSuccess) return fail_rccl(*r, bad_what, first_bad);
  if (res != ncclSuccess) return fail_rccl(*r, "ncclGroupEnd", res);
  if (!self_sendrecv && counts_bytes[root] != 0 && slot_of(root) != d_local) {
    const hipError_t e = hipMemcpyAsync(slot_of(root), d_local, counts_bytes[root], hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return fail(ICAMD_ERR_HIP, "icamd_gather_blocks_rccl: copy of root's own range", e);
  }
  return ICAMD_OK;
} ICAMD_ABI_CATCH

#pragma GCC visibility pop
}  // extern "C"
