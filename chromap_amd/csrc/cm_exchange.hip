// cm_exchange.hip -- multi-GPU record exchange behind the C ABI.
//
// The mapping path shards by read batch and needs no collective.  The reference's final step --
// per chromosome: sort the records, remove PCR duplicates, write (mapping_processor.h:100-202,
// mapping_writer.h:166-376, driven from chromap.h:1305-1355) -- is done by the rank that OWNS the
// chromosome: after a batch is mapped its records are grouped by owner on the device (two-pass
// counting partition), the per-destination counts are all-gathered, one grouped send/recv moves
// every record to its owner (each record crosses xGMI once) straight into the owner's record
// store.  RCCL is called from here, on the context's own mapping stream, so a step is
//   map kernels -> partition kernels -> ncclAllGather(counts) -> ncclSend/ncclRecv -> (next batch)
// on one hot queue; the only host wait is the 8*world*world-byte count matrix.
//
// RCCL is bound at run time, and it has to be the RCCL that sits on the SAME HIP runtime as this
// library: a process may hold two (the system's under /opt/rocm and the one a torch wheel ships;
// which of them this library got depends on load order), and a stream created by one runtime means
// nothing to the other.  So: find the file this library's hipStreamSynchronize comes from, load the librccl next to it.  A
// single-GPU run never touches RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "cm_ctx.h"
#include "cm_kernels.h"

#define EX_BLOCK 256
#define EX_MAX_WORLD 64

#define EXCHECK(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      cm_set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                  \
      return CMGPU_EHIP;                                                                     \
    }                                                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------
// RCCL entry points, resolved once
// ---------------------------------------------------------------------------------------
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err, path;
};
static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static const RcclApi *rccl_api() {
  std::call_once(g_rccl_once, []() {
    std::vector<std::string> names;
    Dl_info di;
    if (dladdr(reinterpret_cast<const void *>(&hipStreamSynchronize), &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t sl = dir.rfind('/');
      dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
      names.push_back(dir + "/librccl.so.1");
      names.push_back(dir + "/librccl.so");
    }
    names.push_back("librccl.so.1");
    for (const std::string &n : names) {
      g_rccl.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) { g_rccl.path = n; break; }
    }
    if (!g_rccl.handle) { g_rccl.err = std::string("cannot load RCCL (librccl.so.1): ") + dlerror(); return; }
#define EX_SYM(field, name)                                                         \
  *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.handle, name);         \
  if (!g_rccl.field) { g_rccl.err = std::string("RCCL lacks ") + name; return; }
    EX_SYM(GetUniqueId, "ncclGetUniqueId")
    EX_SYM(CommInitRank, "ncclCommInitRank")
    EX_SYM(CommInitAll, "ncclCommInitAll")
    EX_SYM(CommDestroy, "ncclCommDestroy")
    EX_SYM(AllGather, "ncclAllGather")
    EX_SYM(Send, "ncclSend")
    EX_SYM(Recv, "ncclRecv")
    EX_SYM(GroupStart, "ncclGroupStart")
    EX_SYM(GroupEnd, "ncclGroupEnd")
    EX_SYM(GetErrorString, "ncclGetErrorString")
#undef EX_SYM
  });
  return g_rccl.err.empty() ? &g_rccl : nullptr;
}

#define NCCLCHECK(ctx, api, call)                                                                        \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess) {                                                                             \
      cm_set_error(ctx, std::string(#call) + ": " + (api)->GetErrorString(r_));                          \
      return CMGPU_EHIP;                                                                                 \
    }                                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------
// owner table: contiguous rid ranges (the output is chromosome-major, so sections concatenate in rank order)
// chosen so that the largest owner's total sequence length B is minimal -- B by bisection on "do the
// sequences, taken in order and packed greedily into bins of B bases, need at most `world` bins", then that
// packing.  GRCh38's chromosomes on 8 ranks: 8.1-15.0 % per rank (rid * world / n_seq gave rank 0 22 %).
// A record's rid is an index rid, or -- with --chr-order -- the rank of its sequence, so the lengths are
// taken in that space.  chromap_amd/distributed.py: owner_table is the twin.
// ---------------------------------------------------------------------------------------
std::vector<uint8_t> cm_owner_table(const cmgpu_ctx *c, uint32_t world) {
  const uint32_t n = c->n_seq;
  std::vector<uint64_t> len(n, 0);
  for (uint32_t i = 0; i < n; ++i) len[c->has_rank && c->h_rank.size() == n ? c->h_rank[i] : i] = c->h_ref_len[i];
  uint64_t total = 0, longest = 0;
  for (uint32_t i = 0; i < n; ++i) { total += len[i]; longest = len[i] > longest ? len[i] : longest; }
  auto bins = [&](uint64_t B) {
    uint64_t p = 1, cur = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (cur + len[i] > B) { ++p; cur = len[i]; } else cur += len[i];
    }
    return p;
  };
  uint64_t lo = longest, hi = total;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (bins(mid) <= world) hi = mid; else lo = mid + 1;
  }
  std::vector<uint8_t> owner(n ? n : 1, 0);
  uint64_t cur = 0;
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (cur + len[i] > lo && cur > 0) { ++k; cur = 0; }
    cur += len[i];
    owner[i] = (uint8_t)(k < world ? k : world - 1);
  }
  return owner;
}

extern "C" int cmgpu_exchange_owner_table(const cmgpu_ctx *c, uint32_t world, uint8_t *owner_out, uint32_t n_sequences) {
  if (!c || !owner_out || world == 0 || world > EX_MAX_WORLD || n_sequences != c->n_seq) return CMGPU_EINVAL;
  const std::vector<uint8_t> t = cm_owner_table(c, world);
  memcpy(owner_out, t.data(), n_sequences);
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// send side: records grouped by owner (order inside a group is irrelevant: a total-order sort follows)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ex_owner(const uint8_t *__restrict__ rec, uint32_t i, const uint8_t *__restrict__ owner, uint32_t n_seq) {
  const uint32_t rid = reinterpret_cast<const uint32_t *>(rec + (uint64_t)i * 24)[1];
  return rid < n_seq ? owner[rid] : 0u;
}
// pass 1: records per owner (block histogram in LDS, one global atomic per owner and block).  A block takes EX_PER_BLOCK records -- a
// block per 256 records was 15 625 blocks for a 4 M-pair batch, each with up to `world` atomics on the same `world` addresses, and
// same-address atomics retire at ~90 per microsecond: 0.17 ms per pass of a 7 ms step spent queueing
#define EX_ITEMS 16
#define EX_PER_BLOCK (EX_BLOCK * EX_ITEMS)
__global__ __launch_bounds__(EX_BLOCK) void k_ex_count(const uint8_t *__restrict__ rec, const uint8_t *__restrict__ ok, uint32_t n,
                                                        const uint8_t *__restrict__ owner, uint32_t n_seq, uint32_t world,
                                                        unsigned long long *__restrict__ counts) {
  __shared__ uint32_t hist[EX_MAX_WORLD];
  if (threadIdx.x < EX_MAX_WORLD) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * EX_PER_BLOCK;
#pragma unroll 4
  for (uint32_t k = 0; k < EX_ITEMS; ++k) {
    const uint32_t i = base + k * EX_BLOCK + threadIdx.x;
    if (i < n && ok[i]) atomicAdd(&hist[ex_owner(rec, i, owner, n_seq)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < world && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}
// the sections' first records: cursors[r] = sum of counts[0..r)
__global__ void k_ex_starts(const unsigned long long *__restrict__ counts, unsigned long long *__restrict__ cursors, uint32_t world) {
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (uint32_t r = 0; r < world; ++r) { cursors[r] = t; t += counts[r]; }
  }
}
// pass 2: every block reserves its share of each owner's section and copies its records there;
// rb = 24: cmgpu_record, rb = 32: cmgpu_record_bc (barcode key appended).  The same EX_PER_BLOCK records per block as pass 1.
__global__ __launch_bounds__(EX_BLOCK) void k_ex_scatter(const uint8_t *__restrict__ rec, const uint8_t *__restrict__ ok,
                                                          const uint64_t *__restrict__ bc, uint32_t n, const uint8_t *__restrict__ owner,
                                                          uint32_t n_seq, uint32_t world, unsigned long long *__restrict__ cursors,
                                                          uint8_t *__restrict__ dst, uint32_t rb, uint32_t per_pair) {
  __shared__ uint32_t hist[EX_MAX_WORLD];
  __shared__ unsigned long long base[EX_MAX_WORLD];
  if (threadIdx.x < EX_MAX_WORLD) hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * EX_PER_BLOCK;
  uint32_t own[EX_ITEMS], local[EX_ITEMS];
#pragma unroll
  for (uint32_t k = 0; k < EX_ITEMS; ++k) {
    const uint32_t i = b0 + k * EX_BLOCK + threadIdx.x;
    own[k] = 0xffffffffu; local[k] = 0;
    if (i < n && ok[i]) { own[k] = ex_owner(rec, i, owner, n_seq); local[k] = atomicAdd(&hist[own[k]], 1u); }
  }
  __syncthreads();
  if (threadIdx.x < world && hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < EX_ITEMS; ++k) {
    if (own[k] == 0xffffffffu) continue;
    const uint32_t i = b0 + k * EX_BLOCK + threadIdx.x;
    const uint64_t *sp = reinterpret_cast<const uint64_t *>(rec + (uint64_t)i * 24);
    uint64_t *dp = reinterpret_cast<uint64_t *>(dst + (base[own[k]] + local[k]) * rb);
    dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2];
    if (rb == 32) dp[3] = bc[i / per_pair];
  }
}

// this rank's own records go from the send buffer to the store by a plain copy kernel (a device-to-device hipMemcpyAsync of
// ~100 MB took 1.4 ms here -- the copy engines' rate; the CUs move it in well under 0.1 ms)
__global__ __launch_bounds__(EX_BLOCK) void k_ex_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t bytes) {
  const uint64_t n8 = bytes >> 3;  // records are 24 or 32 bytes: both ends 8-byte aligned
  const uint64_t *s8 = reinterpret_cast<const uint64_t *>(src);
  uint64_t *d8 = reinterpret_cast<uint64_t *>(dst);
  for (uint64_t i = (uint64_t)blockIdx.x * EX_BLOCK + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * EX_BLOCK) d8[i] = s8[i];
}

static int ex_owner_upload(cmgpu_ctx *c, uint32_t world) {
  CmExchange &x = c->ex;
  x.h_owner = cm_owner_table(c, world);
  if (x.owner.ensure(x.h_owner.size() + 16)) { cm_set_error(c, "out of device memory (owner table)"); return CMGPU_ENOMEM; }
  EXCHECK(c, hipMemcpy(x.owner.p, x.h_owner.data(), x.h_owner.size(), hipMemcpyHostToDevice));
  return CMGPU_OK;
}

// groups the resident batch's records by owner into dst (device); counts on the device at d_counts[0..world)
static int ex_partition(cmgpu_ctx *c, uint32_t n, uint32_t world, const uint8_t *d_owner, uint8_t *dst, uint32_t rb,
                        unsigned long long *d_counts, unsigned long long *d_cursors) {
  hipStream_t s = c->stream;
  EXCHECK(c, hipMemsetAsync(d_counts, 0, (EX_MAX_WORLD + 1) * 8, s));
  if (n) {
    const dim3 g((n + EX_PER_BLOCK - 1) / EX_PER_BLOCK), b(EX_BLOCK);
    hipLaunchKernelGGL(k_ex_count, g, b, 0, s, (const uint8_t *)c->rec.p, (const uint8_t *)c->rec_ok.p, n, d_owner, c->n_seq, world, d_counts);
    hipLaunchKernelGGL(k_ex_starts, dim3(1), dim3(64), 0, s, d_counts, d_cursors, world);
    hipLaunchKernelGGL(k_ex_scatter, g, b, 0, s, (const uint8_t *)c->rec.p, (const uint8_t *)c->rec_ok.p,
                       rb == 32 ? (const uint64_t *)c->bc_key.p : (const uint64_t *)nullptr, n, d_owner, c->n_seq, world, d_cursors, dst, rb, cm_rec_per_pair(c));
  }
  return CMGPU_OK;
}

// cmgpu_records_partition (declared next to cmgpu_records_to_device): the send buffer of an exchange the
// caller performs itself; same owner rule and kernels as cmgpu_exchange_step
extern "C" int cmgpu_records_partition(cmgpu_ctx *c, uint32_t world, void *device_dst, uint64_t capacity, uint64_t *counts) {
  if (!c || !device_dst || !counts || world == 0 || world > EX_MAX_WORLD) return CMGPU_EINVAL;
  EXCHECK(c, cm_enter(c));
  for (uint32_t r = 0; r < world; ++r) counts[r] = 0;
  if (c->n_pairs == 0) return CMGPU_OK;
  if (capacity < cm_rec_slots(c)) { cm_set_error(c, "send buffer too small (one slot per pair of the batch -- times max_num_best_mappings -- is needed)"); return CMGPU_ECAPACITY; }
  if (cm_pairs_records(c)) { cm_set_error(c, "pairs records are not partitioned by chromosome"); return CMGPU_EINVAL; }
  const std::vector<uint8_t> t = cm_owner_table(c, world);
  DevBuf &dcnt = c->part_cnt;
  if (dcnt.ensure(2 * (EX_MAX_WORLD + 1) * 8 + t.size() + 16)) { cm_set_error(c, "out of device memory (partition)"); return CMGPU_ENOMEM; }
  unsigned long long *d_counts = (unsigned long long *)dcnt.p, *d_cursors = d_counts + EX_MAX_WORLD + 1;
  uint8_t *d_owner = (uint8_t *)(d_cursors + EX_MAX_WORLD + 1);
  EXCHECK(c, hipMemcpyAsync(d_owner, t.data(), t.size(), hipMemcpyHostToDevice, c->stream));
  int rc = ex_partition(c, (uint32_t)cm_rec_slots(c), world, d_owner, (uint8_t *)device_dst, 24, d_counts, d_cursors);
  if (rc) return rc;
  unsigned long long h[EX_MAX_WORLD];
  EXCHECK(c, hipMemcpyAsync(h, d_counts, (size_t)world * 8, hipMemcpyDeviceToHost, c->stream));
  EXCHECK(c, cm_stream_sync(c->stream));
  for (uint32_t r = 0; r < world; ++r) counts[r] = h[r];
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// set-up / tear-down
// ---------------------------------------------------------------------------------------
static int ex_common_init(cmgpu_ctx *c, int rank, int world) {
  if (!c || world < 1 || world > EX_MAX_WORLD || rank < 0 || rank >= world) { cm_set_error(c, "bad rank / world"); return CMGPU_EINVAL; }
  if (c->ex.transport != 0) { cm_set_error(c, "the exchange is already initialised"); return CMGPU_EINVAL; }
  if (cm_pairs_records(c)) { cm_set_error(c, "pairs records are post-processed by one rank; no chromosome owners"); return CMGPU_EINVAL; }
  EXCHECK(c, cm_enter(c));
  CmExchange &x = c->ex;
  x.rank = rank;
  x.world = world;
  x.sent_total = x.recv_total = x.steps = 0;
  for (uint64_t &v : x.owned_by) v = 0;
  int rc = ex_owner_upload(c, (uint32_t)world);
  if (rc) return rc;
  if (x.counts.ensure((2 * (EX_MAX_WORLD + 1) + EX_MAX_WORLD * (EX_MAX_WORLD + 1)) * 8)) { cm_set_error(c, "out of device memory (exchange counts)"); return CMGPU_ENOMEM; }
  if (!x.h_matrix) EXCHECK(c, hipHostMalloc((void **)&x.h_matrix, EX_MAX_WORLD * (EX_MAX_WORLD + 1) * 8, hipHostMallocDefault));
  if (!x.stream) {
    EXCHECK(c, hipStreamCreate(&x.stream));
    EXCHECK(c, hipEventCreateWithFlags(&x.ev_part, hipEventDisableTiming));
    EXCHECK(c, hipEventCreateWithFlags(&x.ev_payload, hipEventDisableTiming));
  }
  x.payload_pending = false;
  return CMGPU_OK;
}

extern "C" int cmgpu_exchange_unique_id(void *id_out) {
  if (!id_out) return CMGPU_EINVAL;
  const RcclApi *api = rccl_api();
  if (!api) { cm_set_error(nullptr, g_rccl.err); return CMGPU_EHIP; }
  ncclUniqueId id;
  const ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) { cm_set_error(nullptr, std::string("ncclGetUniqueId: ") + api->GetErrorString(r)); return CMGPU_EHIP; }
  static_assert(sizeof(id) == CMGPU_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return CMGPU_OK;
}

extern "C" int cmgpu_exchange_init(cmgpu_ctx *c, const void *unique_id, int rank, int world) {
  if (!c || !unique_id) return CMGPU_EINVAL;
  const RcclApi *api = rccl_api();
  if (!api) { cm_set_error(c, g_rccl.err); return CMGPU_EHIP; }
  int rc = ex_common_init(c, rank, world);
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  NCCLCHECK(c, api, api->CommInitRank(&comm, world, id, rank));
  c->ex.comm = comm;
  c->ex.transport = 1;
  return CMGPU_OK;
}

extern "C" int cmgpu_exchange_init_all(cmgpu_ctx *const *ctxs, int n) {
  if (!ctxs || n < 1 || n > EX_MAX_WORLD) return CMGPU_EINVAL;
  const RcclApi *api = rccl_api();
  if (!api) { cm_set_error(ctxs[0], g_rccl.err); return CMGPU_EHIP; }
  std::vector<int> dev(n);
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return CMGPU_EINVAL;
    dev[i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      if (dev[j] == dev[i]) { cm_set_error(ctxs[0], "cmgpu_exchange_init_all: two contexts on one device (RCCL needs one rank per GPU)"); return CMGPU_EINVAL; }
  }
  for (int i = 0; i < n; ++i) { const int rc = ex_common_init(ctxs[i], i, n); if (rc) return rc; }
  std::vector<ncclComm_t> comms(n, nullptr);
  NCCLCHECK(ctxs[0], api, api->CommInitAll(comms.data(), n, dev.data()));
  for (int i = 0; i < n; ++i) { ctxs[i]->ex.comm = comms[i]; ctxs[i]->ex.transport = 1; }
  return CMGPU_OK;
}

extern "C" int cmgpu_exchange_init_external(cmgpu_ctx *c, const cmgpu_exchange_transport *t, int rank, int world) {
  if (!c || !t || !t->allgather_counts || !t->alltoallv) return CMGPU_EINVAL;
  int rc = ex_common_init(c, rank, world);
  if (rc) return rc;
  c->ex.ext = *t;
  c->ex.transport = 2;
  return CMGPU_OK;
}

int cm_exchange_quiesce(cmgpu_ctx *c) {
  CmExchange &x = c->ex;
  if (x.stream && x.payload_pending) {
    (void)hipSetDevice(c->device);
    const hipError_t e = hipStreamSynchronize(x.stream);
    x.payload_pending = false;
    if (e != hipSuccess) { cm_set_error(c, std::string("exchange payload: ") + hipGetErrorString(e)); return CMGPU_EHIP; }
  }
  return CMGPU_OK;
}

void cm_exchange_release(cmgpu_ctx *c) {
  CmExchange &x = c->ex;
  (void)cm_exchange_quiesce(c);
  if (x.ev_part) { (void)hipEventDestroy(x.ev_part); x.ev_part = nullptr; }
  if (x.ev_payload) { (void)hipEventDestroy(x.ev_payload); x.ev_payload = nullptr; }
  if (x.stream) { (void)hipStreamDestroy(x.stream); x.stream = nullptr; }
  if (x.comm) {
    const RcclApi *api = rccl_api();
    if (api) (void)api->CommDestroy((ncclComm_t)x.comm);
    x.comm = nullptr;
  }
  if (x.h_matrix) { (void)hipHostFree(x.h_matrix); x.h_matrix = nullptr; }
  x.transport = 0;
}

extern "C" int cmgpu_exchange_finalize(cmgpu_ctx *c) {
  if (!c) return CMGPU_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)cm_exchange_quiesce(c);
  cm_exchange_release(c);
  return CMGPU_OK;
}

// plain copies between host and device memory through the library's own HIP runtime (a host-staged exchange
// transport must not bring a second runtime into the process); kind 1: host -> device, 2: device -> host
extern "C" int cmgpu_memcpy(cmgpu_ctx *c, void *dst, const void *src, uint64_t bytes, int kind) {
  if (!c || (bytes && (!dst || !src)) || (kind != 1 && kind != 2)) return CMGPU_EINVAL;
  EXCHECK(c, cm_enter(c));
  if (bytes) EXCHECK(c, hipMemcpy(dst, src, bytes, kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

extern "C" int cmgpu_exchange_info(const cmgpu_ctx *c, int *rank, int *world, uint64_t *records_sent, uint64_t *records_received) {
  if (!c) return CMGPU_EINVAL;
  if (rank) *rank = c->ex.rank;
  if (world) *world = c->ex.transport ? c->ex.world : 0;
  if (records_sent) *records_sent = c->ex.sent_total;
  if (records_received) *records_received = c->ex.recv_total;
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// The communication plan of one round for rank `me`, from the count matrix every rank holds after the first all-gather
// (M[r * stride + j] = records rank r sends to rank j).  What cmgpu_exchange_step posts, in this order, and nothing else:
//   CM_EX_ALLGATHER_COUNTS, CM_EX_ALLGATHER_STATUS   unconditional (world > 1 for the status word) -- an empty rank takes part;
//   CM_EX_COPY_SELF                                  its own records (no peer involved);
//   one group of CM_EX_SEND / CM_EX_RECV             for d = 1 .. world - 1: send to me + d, receive from me - d (staggered:
//                                                    no rank is everybody's first target), a pair only when ITS matrix entry is
//                                                    non-zero -- the sender reads M[me][peer], the receiver the same entry
//                                                    M[src][me], so a Send is posted iff its Recv is.
// A pure function of (me, world, M): tests/test_exchange_plan.py replays it for every rank of a world on the CPU and checks
// that the rounds cannot deadlock (same collectives on every rank, every Send met by one Recv of the same size in the same group).
// ---------------------------------------------------------------------------------------
std::vector<cmgpu_exchange_op> cm_exchange_plan(uint32_t me, uint32_t world, const uint64_t *M, uint32_t stride) {
  std::vector<cmgpu_exchange_op> ops;
  ops.push_back({CMGPU_EX_ALLGATHER_COUNTS, 0, (uint64_t)stride});
  if (world > 1) ops.push_back({CMGPU_EX_ALLGATHER_STATUS, 0, 1});
  if (M[(size_t)me * stride + me]) ops.push_back({CMGPU_EX_COPY_SELF, me, M[(size_t)me * stride + me]});
  for (uint32_t d = 1; d < world; ++d) {
    const uint32_t peer = (me + d) % world, src = (me + world - d) % world;
    if (M[(size_t)me * stride + peer]) ops.push_back({CMGPU_EX_SEND, peer, M[(size_t)me * stride + peer]});
    if (M[(size_t)src * stride + me]) ops.push_back({CMGPU_EX_RECV, src, M[(size_t)src * stride + me]});
  }
  return ops;
}
extern "C" int cmgpu_exchange_plan(uint32_t rank, uint32_t world, const uint64_t *matrix, cmgpu_exchange_op *ops, uint32_t capacity, uint32_t *n_ops) {
  if (!matrix || !n_ops || world == 0 || world > EX_MAX_WORLD || rank >= world) return CMGPU_EINVAL;
  const std::vector<cmgpu_exchange_op> p = cm_exchange_plan(rank, world, matrix, world);
  *n_ops = (uint32_t)p.size();
  if (p.size() > capacity) return ops ? CMGPU_ECAPACITY : CMGPU_OK;
  for (size_t i = 0; ops && i < p.size(); ++i) ops[i] = p[i];
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// one round
// ---------------------------------------------------------------------------------------
extern "C" int cmgpu_exchange_step(cmgpu_ctx *c, uint64_t *sent_per_rank, uint64_t *n_received) {
  if (!c) return CMGPU_EINVAL;
  CmExchange &x = c->ex;
  if (x.transport == 0) { cm_set_error(c, "the exchange is not initialised (cmgpu_exchange_init*)"); return CMGPU_EINVAL; }
  EXCHECK(c, cm_enter(c));
  hipStream_t s = c->stream;
  // a batch takes part once: a second step without a new cmgpu_map_* call contributes an empty batch (a rank that
  // ran out of input keeps calling while the others finish)
  const uint32_t world = (uint32_t)x.world, me = (uint32_t)x.rank, n = c->batch_exchanged ? 0 : (uint32_t)cm_rec_slots(c), stride = world + 1;
  // record kind of this rank: 24-byte bulk records or 32-byte {record, barcode}; 0 = nothing to say (empty batch,
  // empty store).  It travels as entry `world` of the count vector so that every rank uses the same size.
  const uint32_t rb_local = n ? (c->has_barcodes ? 32u : 24u) : (c->store_n ? (c->store_has_bc ? 32u : 24u) : 0u);
  if (n_received) *n_received = 0;
  if (x.send.cap < (size_t)(n ? n : 1) * 32 + 16) {  // growing the send buffer frees the old one: no payload may be reading it
    int qrc = cm_exchange_quiesce(c);
    if (qrc) return qrc;
  }
  if (x.send.ensure((size_t)(n ? n : 1) * 32 + 16)) { cm_set_error(c, "out of device memory (exchange send buffer)"); return CMGPU_ENOMEM; }
  // the previous round's payload still reads the send buffer (and it was issued on the payload stream)
  if (x.payload_pending) EXCHECK(c, hipStreamWaitEvent(s, x.ev_payload, 0));
  unsigned long long *d_counts = (unsigned long long *)x.counts.p, *d_cursors = d_counts + EX_MAX_WORLD + 1, *d_matrix = d_cursors + EX_MAX_WORLD + 1;
  int rc = ex_partition(c, n, world, (const uint8_t *)x.owner.p, (uint8_t *)x.send.p, rb_local ? rb_local : 24u, d_counts, d_cursors);
  if (rc) return rc;
  // ---- counts: row r of the matrix = what rank r sends to everyone (+ its record kind)
  unsigned long long *M = x.h_matrix;
  M[0] = rb_local;
  EXCHECK(c, hipMemcpyAsync(d_counts + world, M, 8, hipMemcpyHostToDevice, s));
  if (x.transport == 1) {
    const RcclApi *api = rccl_api();
    NCCLCHECK(c, api, api->AllGather(d_counts, d_matrix, stride, ncclUint64, (ncclComm_t)x.comm, s));
    EXCHECK(c, hipMemcpyAsync(M, d_matrix, (size_t)world * stride * 8, hipMemcpyDeviceToHost, s));
    EXCHECK(c, cm_stream_sync(s));
  } else {
    uint64_t mine[EX_MAX_WORLD + 1];
    EXCHECK(c, hipMemcpyAsync(M, d_counts, (size_t)stride * 8, hipMemcpyDeviceToHost, s));
    EXCHECK(c, cm_stream_sync(s));
    for (uint32_t r = 0; r < stride; ++r) mine[r] = M[r];
    std::vector<uint64_t> full((size_t)world * stride, 0);
    if (x.ext.allgather_counts(x.ext.user, mine, full.data(), stride) != 0) { cm_set_error(c, "exchange transport: allgather_counts failed"); return CMGPU_EIO; }
    for (size_t i = 0; i < full.size(); ++i) M[i] = full[i];
  }
  uint32_t rb = rb_local;
  for (uint32_t r = 0; r < world; ++r) {
    const uint32_t k = (uint32_t)M[(size_t)r * stride + world];
    if (k == 0) continue;
    if (rb == 0) rb = k;
    if (k != rb || (k != 24 && k != 32)) { cm_set_error(c, "the ranks disagree on the record kind (bulk / single-cell)"); return CMGPU_EINVAL; }
  }
  if (rb == 0) rb = 24;
  const bool bc = rb == 32;
  const std::vector<uint64_t> mat(M, M + (size_t)world * stride);  // (the status round below reuses M)
  uint64_t send_cnt[EX_MAX_WORLD], recv_cnt[EX_MAX_WORLD], send_off[EX_MAX_WORLD], recv_off[EX_MAX_WORLD], tot_s = 0, tot_r = 0;
  for (uint32_t r = 0; r < world; ++r) {
    send_cnt[r] = M[(size_t)me * stride + r];
    recv_cnt[r] = M[(size_t)r * stride + me];
    send_off[r] = tot_s; tot_s += send_cnt[r];
    recv_off[r] = tot_r; tot_r += recv_cnt[r];
    if (sent_per_rank) sent_per_rank[r] = send_cnt[r];
  }
  for (uint32_t j = 0; j < world; ++j)
    for (uint32_t r = 0; r < world; ++r) x.owned_by[j] += M[(size_t)r * stride + j];
  // Everything that can fail on THIS rank alone (store kind, store / staging growth) happens before any payload is posted,
  // and its outcome is agreed on by all ranks: a rank that returned early here while its peers posted the matching
  // Send / Recv would leave them waiting in RCCL forever.
  int local_rc = CMGPU_OK;
  uint8_t *dest = nullptr;
  if (c->store_n && tot_r && c->store_has_bc != bc) { cm_set_error(c, "record store mixes barcoded and bulk batches"); local_rc = CMGPU_EINVAL; }
  if (!local_rc && tot_r) {
    if (c->store_n + tot_r > c->store_cap || (bc && !c->store_bc.p)) local_rc = cm_exchange_quiesce(c);  // the store moves: the previous payload must have landed
    if (!local_rc) local_rc = cm_store_reserve(c, c->store_n + tot_r, bc);
    if (!local_rc && bc) {
      if (x.stage.cap < (size_t)tot_r * 32 + 16) local_rc = cm_exchange_quiesce(c);
      if (!local_rc && x.stage.ensure((size_t)tot_r * 32 + 16)) { cm_set_error(c, "out of device memory (exchange staging)"); local_rc = CMGPU_ENOMEM; }
      if (!local_rc) dest = (uint8_t *)x.stage.p;
    } else if (!local_rc) {
      dest = (uint8_t *)c->store.p + c->store_n * 24;
    }
  }
  if (world > 1) {  // one status word per rank; any failure makes every rank leave this step together
    uint64_t status[EX_MAX_WORLD];
    const uint64_t mine = (uint64_t)(uint32_t)(-local_rc);
    if (x.transport == 1) {
      const RcclApi *api = rccl_api();
      M[0] = mine;
      EXCHECK(c, hipMemcpyAsync(d_counts, M, 8, hipMemcpyHostToDevice, s));
      NCCLCHECK(c, api, api->AllGather(d_counts, d_matrix, 1, ncclUint64, (ncclComm_t)x.comm, s));
      EXCHECK(c, hipMemcpyAsync(M, d_matrix, (size_t)world * 8, hipMemcpyDeviceToHost, s));
      EXCHECK(c, cm_stream_sync(s));
      for (uint32_t r = 0; r < world; ++r) status[r] = M[r];
    } else {
      uint64_t all[EX_MAX_WORLD * 2];
      if (x.ext.allgather_counts(x.ext.user, &mine, all, 1) != 0) { cm_set_error(c, "exchange transport: allgather_counts failed"); return CMGPU_EIO; }
      for (uint32_t r = 0; r < world; ++r) status[r] = all[r];
    }
    if (local_rc) return local_rc;
    for (uint32_t r = 0; r < world; ++r)
      if (status[r]) { cm_set_error(c, "exchange: rank " + std::to_string(r) + " failed before the payload (status -" + std::to_string((unsigned long long)status[r]) + ")"); return CMGPU_EIO; }
  } else if (local_rc) {
    return local_rc;
  }
  if (x.transport == 1) {
    // The payload: this rank's own records by a device-to-device copy, the peers' by one grouped send / recv.  By default on
    // the mapping stream; with the "exchange_overlap" option on a stream of its own, under the next batch's kernels (the
    // send buffer and the store's tail are its only operands; both are guarded above and in cm_exchange_quiesce) --
    // measured slower on one GPU with several lanes mapping beside it, so it is not the default.
    const RcclApi *api = rccl_api();
    const bool overlap = c->opt_exchange_overlap != 0;
    hipStream_t ps = overlap ? x.stream : s;
    if (overlap) {
      EXCHECK(c, hipEventRecord(x.ev_part, s));
      EXCHECK(c, hipStreamWaitEvent(ps, x.ev_part, 0));
    }
    // (the posts follow cm_exchange_plan: the all-gathers above were its first entries)
    const std::vector<cmgpu_exchange_op> plan = cm_exchange_plan(me, world, mat.data(), stride);
    bool in_group = false;
    for (const cmgpu_exchange_op &op : plan) {
      if (op.kind == CMGPU_EX_COPY_SELF) {
        const uint64_t bytes = op.records * rb;
        uint64_t blocks = (bytes / 8 + EX_BLOCK * 8 - 1) / (EX_BLOCK * 8);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_ex_copy, dim3((unsigned)blocks), dim3(EX_BLOCK), 0, ps, (const uint8_t *)x.send.p + send_off[me] * rb, dest + recv_off[me] * rb, bytes);
      } else if (op.kind == CMGPU_EX_SEND || op.kind == CMGPU_EX_RECV) {
        if (!in_group) { NCCLCHECK(c, api, api->GroupStart()); in_group = true; }
        if (op.kind == CMGPU_EX_SEND) NCCLCHECK(c, api, api->Send((const uint8_t *)x.send.p + send_off[op.peer] * rb, op.records * rb, ncclUint8, (int)op.peer, (ncclComm_t)x.comm, ps));
        else NCCLCHECK(c, api, api->Recv(dest + recv_off[op.peer] * rb, op.records * rb, ncclUint8, (int)op.peer, (ncclComm_t)x.comm, ps));
      }
    }
    if (in_group) NCCLCHECK(c, api, api->GroupEnd());
    if (tot_r && bc) cm_store_split_bc(c, dest, tot_r, ps);
    if (overlap) {
      EXCHECK(c, hipEventRecord(x.ev_payload, ps));
      x.payload_pending = true;
    }
  } else {
    // the callbacks run outside the stream: the send buffer is complete (synchronised above)
    if (x.ext.alltoallv(x.ext.user, x.send.p, send_cnt, dest, recv_cnt, world, rb) != 0) { cm_set_error(c, "exchange transport: alltoallv failed"); return CMGPU_EIO; }
  }
  if (x.transport != 1 && tot_r && bc) cm_store_split_bc(c, dest, tot_r, s);
  c->store_n += tot_r;
  c->batch_exchanged = true;
  x.sent_total += tot_s;
  x.recv_total += tot_r;
  ++x.steps;
  if (n_received) *n_received = tot_r;
  return CMGPU_OK;
}

// the device code of this translation unit is loaded by the HIP runtime at the first launch of one of its kernels (milliseconds to tens of
// milliseconds for the larger ones): context creation launches this empty kernel so that a job's first batch does not pay for it (cm_api.hip: cm_load_device_code)
__global__ void k_touch_exchange() {}
void cm_touch_exchange(hipStream_t s) { hipLaunchKernelGGL(k_touch_exchange, dim3(1), dim3(1), 0, s); }
