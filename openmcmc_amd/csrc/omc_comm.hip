// The one collective of the path: gather of the per-rank sample stores on a root rank (RCCL over xGMI).
//
// Chains shard over GPUs and never interact; after the run every rank holds
// store[n_outer][C_rank][row] and the root wants store[n_outer][C_total][row] with the chains in rank
// order = global chain order (what MCMC.store holds in the reference for one chain, mcmc.py:105-111, for
// all of them).  xGMI is point to point -- every peer has its own link into the root -- so all peers send
// at once (one ncclSend/ncclRecv pair per peer inside one group, no ring), each its contiguous block of a
// slab of outer indices; the root receives into a bounded staging buffer and a copy kernel interleaves
// the blocks into place (one extra pass at HBM speed against a transfer at link speed).  Shards may be
// uneven; n_outer == 1 or a single peer block per outer index needs no staging at all.
#include <rccl/rccl.h>
#include <string.h>

#include <string>
#include <vector>

#include "omc_common.h"

struct omc_comm {
  ncclComm_t comm;
  int world, rank, device;
  double* staging;
  size_t staging_bytes;
};

static omc_status nccl_fail(const char* what, ncclResult_t r) {
  omc_set_error_text((std::string(what) + ": " + ncclGetErrorString(r)).c_str());
  return OMC_HIP_ERROR;
}
#define OMC_NCCL_CHECK(expr)                        \
  do {                                              \
    ncclResult_t _r = (expr);                       \
    if (_r != ncclSuccess) return nccl_fail(#expr, _r); \
  } while (0)

// dst[o][off + c][e] = src[o][c][e]  for o < n_outer, c < cr, e < row   (src contiguous [n_outer][cr][row])
__global__ void __launch_bounds__(256) k_interleave_block(const double* __restrict__ src, double* __restrict__ dst, int64_t n_outer,
                                                          int64_t cr_row, int64_t dst_stride, int64_t dst_off) {
  const int64_t total = n_outer * cr_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = i / cr_row, r = i - o * cr_row;
    dst[o * dst_stride + dst_off + r] = src[i];
  }
}

static void launch_interleave(hipStream_t st, const double* src, double* dst, int64_t n_outer, int64_t cr_row, int64_t dst_stride,
                              int64_t dst_off) {
  const int64_t total = n_outer * cr_row;
  if (total <= 0) return;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(k_interleave_block, dim3((unsigned)grid), dim3(256), 0, st, src, dst, n_outer, cr_row, dst_stride, dst_off);
}

extern "C" {

omc_status omc_comm_unique_id(char* id_out, int64_t id_bytes) {
  if (!id_out || id_bytes < (int64_t)sizeof(ncclUniqueId)) return OMC_INVALID_ARG;
  ncclUniqueId id;
  OMC_NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return OMC_OK;
}

int64_t omc_comm_unique_id_bytes(void) { return (int64_t)sizeof(ncclUniqueId); }

omc_status omc_comm_create(omc_ctx* ctx, int32_t world, int32_t rank, const char* id, int64_t id_bytes, omc_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id || id_bytes < (int64_t)sizeof(ncclUniqueId)) return OMC_INVALID_ARG;
  *out = nullptr;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  omc_comm* c = new omc_comm();
  c->world = world; c->rank = rank; c->device = ctx->device;
  c->staging = nullptr; c->staging_bytes = 0;
  ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
  *out = c;
  return OMC_OK;
}

omc_status omc_comm_destroy(omc_comm* comm) {
  if (!comm) return OMC_INVALID_ARG;
  hipSetDevice(comm->device);
  if (comm->staging) hipFree(comm->staging);
  ncclCommDestroy(comm->comm);
  delete comm;
  return OMC_OK;
}

// What moves a peer's block to the root: RCCL in production, device-to-device copies in the single-process stand-in.
struct GatherTransport {
  ncclComm_t comm;             // RCCL: the communicator; loopback: unused
  const double* const* sends;  // loopback: the send buffer of every rank (device pointers, all on this GPU); RCCL: NULL
};

// The gather as rank `me` of W sees it (see the head of the file).  Root side: receive the peers' blocks of a slab of outer
// indices -- straight into place when there is one outer index, else into the staging buffer in rank order -- then
// interleave the staged blocks into recv[o][chain offset of the peer + c][e].
static omc_status gather_core(omc_ctx* ctx, const GatherTransport& tr, int W, int me, int root, const double* send, int64_t n_outer,
                              int64_t row, const int64_t* counts, double* recv, int64_t staging_limit_bytes, double** staging,
                              size_t* staging_bytes) {
  int64_t total = 0, off_me = 0;
  for (int r = 0; r < W; ++r) {
    if (counts[r] < 0) return OMC_INVALID_ARG;
    if (r < me) off_me += counts[r];
    total += counts[r];
  }
  if (counts[me] > 0 && !send) return OMC_INVALID_ARG;
  if (me == root && !recv && total > 0) return OMC_INVALID_ARG;
  OMC_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (n_outer == 0 || total == 0) return OMC_OK;
  const int64_t dst_stride = total * row;

  // the root's own block goes straight into place
  if (me == root) launch_interleave(st, send, recv, n_outer, counts[me] * row, dst_stride, off_me * row);
  if (W == 1) { OMC_HIP_CHECK(hipGetLastError()); return OMC_OK; }

  // slab of outer indices per round: everything when blocks land in place without staging (n_outer == 1),
  // else bounded by the staging budget (default 1 GiB) on the root
  const bool direct = (n_outer == 1);
  int64_t slab = n_outer;
  if (!direct) {
    const int64_t limit = staging_limit_bytes > 0 ? staging_limit_bytes : ((int64_t)1 << 30);
    const int64_t per_outer = (total - counts[root]) * row * (int64_t)sizeof(double);
    slab = per_outer > 0 ? limit / per_outer : n_outer;
    if (slab < 1) slab = 1;
    if (slab > n_outer) slab = n_outer;
    if (me == root) {
      const size_t need = (size_t)slab * (size_t)per_outer;
      if (*staging_bytes < need) {
        OMC_HIP_CHECK(hipStreamSynchronize(st));
        if (*staging) OMC_HIP_CHECK(hipFree(*staging));
        *staging = nullptr; *staging_bytes = 0;
        OMC_HIP_CHECK(hipMalloc(staging, need));
        *staging_bytes = need;
      }
    }
  }
  const bool rccl = tr.sends == nullptr;
  for (int64_t o0 = 0; o0 < n_outer; o0 += slab) {
    const int64_t k = (n_outer - o0 < slab) ? n_outer - o0 : slab;
    if (rccl) OMC_NCCL_CHECK(ncclGroupStart());
    if (me == root) {
      int64_t soff = 0, coff = 0;
      for (int r = 0; r < W; ++r) {
        if (r != root && counts[r] > 0) {
          double* dst = direct ? recv + coff * row : *staging + soff;
          const size_t cnt = (size_t)(k * counts[r] * row);
          if (rccl) {
            OMC_NCCL_CHECK(ncclRecv(dst, cnt, ncclDouble, r, tr.comm, st));
          } else {  // what rank r's ncclSend below would have put on the wire
            OMC_HIP_CHECK(hipMemcpyAsync(dst, tr.sends[r] + o0 * counts[r] * row, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
          }
          soff += k * counts[r] * row;
        }
        coff += counts[r];
      }
    } else if (counts[me] > 0 && rccl) {
      OMC_NCCL_CHECK(ncclSend(send + o0 * counts[me] * row, (size_t)(k * counts[me] * row), ncclDouble, root, tr.comm, st));
    }
    if (rccl) OMC_NCCL_CHECK(ncclGroupEnd());
    if (me == root && !direct) {
      int64_t soff = 0, coff = 0;
      for (int r = 0; r < W; ++r) {
        if (r != root && counts[r] > 0) {
          launch_interleave(st, *staging + soff, recv + o0 * dst_stride, k, counts[r] * row, dst_stride, coff * row);
          soff += k * counts[r] * row;
        }
        coff += counts[r];
      }
    }
  }
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

omc_status omc_gather_samples(omc_ctx* ctx, omc_comm* comm, const double* send, int64_t n_outer, int64_t row,
                              const int64_t* counts, double* recv, int32_t root, int64_t staging_limit_bytes) {
  if (!ctx || !comm || !counts || n_outer < 0 || row < 1 || root < 0 || root >= comm->world) return OMC_INVALID_ARG;
  const GatherTransport tr{comm->comm, nullptr};
  return gather_core(ctx, tr, comm->world, comm->rank, root, send, n_outer, row, counts, recv, staging_limit_bytes, &comm->staging,
                     &comm->staging_bytes);
}

omc_status omc_gather_samples_local(omc_ctx* ctx, int32_t world, const double* const* sends, int64_t n_outer, int64_t row,
                                    const int64_t* counts, double* recv, int32_t root, int64_t staging_limit_bytes) {
  if (!ctx || !sends || !counts || world < 1 || n_outer < 0 || row < 1 || root < 0 || root >= world) return OMC_INVALID_ARG;
  for (int r = 0; r < world; ++r)
    if (counts[r] > 0 && !sends[r]) return OMC_INVALID_ARG;
  const GatherTransport tr{nullptr, sends};
  double* staging = nullptr;
  size_t staging_bytes = 0;
  omc_status st = gather_core(ctx, tr, world, root, root, sends[root], n_outer, row, counts, recv, staging_limit_bytes, &staging,
                              &staging_bytes);
  if (staging) {  // the stand-in owns its staging buffer for the length of the call
    hipStreamSynchronize(ctx->stream);
    hipFree(staging);
  }
  return st;
}

}  // extern "C"
