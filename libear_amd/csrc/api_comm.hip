// api_comm.hip — (J) the one exchange step of the multi-GPU render of include/earhip.h: objects are
// sharded over the GPUs of a node, every GPU renders its shard COMPLETELY (the chain after the buses is
// linear and per channel), and the partial loudspeaker outputs are summed by one RCCL reduce-scatter
// over the channel axis on the context's stream: every rank ends up owning a slice of the channels of the
// shared bus.  libear has no such step (it is single-threaded, single-device): nothing to cite.  The
// process-group part (who is rank r, how the 128-byte id reaches every rank) stays with the caller —
// MPI, torch.distributed, a file: this library only needs the id.
#include <rccl/rccl.h>

#include <cstring>
#include <memory>

#include "common.h"

using namespace earhip;

#define EARHIP_NCCL(expr)                                                                       \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      throw ::earhip::Error{EARHIP_DEVICE_ERROR,                                                \
                            std::string("internal error: RCCL: ") + ncclGetErrorString(r_) + " in " #expr}; \
  } while (0)

struct earhip_comm {
  earhip_ctx *ctx;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // the collective runs on its own stream, ordered against the context's stream by events, so that it
  // overlaps the next render: ready (render of this buffer done), done[slot] (exchange of slot done)
  hipStream_t xstream = nullptr;
  hipEvent_t ready = nullptr, done[2] = {nullptr, nullptr};
  bool issued[2] = {false, false};
  // timing of the collectives of a slot (events on the communicator's stream): [slot][begin, end]
  hipEvent_t tev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  bool timed[2] = {false, false};  // some interval has been recorded
  bool open_[2] = {false, false};  // an exchange opened the interval and no gather has closed it yet
};

extern "C" {

int earhip_comm_unique_id(void *id128) {
  return guarded([&] {
    require(id128 != nullptr, "id must not be NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id around as 128 bytes");
    ncclUniqueId id;
    EARHIP_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
  });
}

int earhip_comm_create(earhip_ctx *ctx, int rank, int world, const void *id128, earhip_comm **out) {
  return guarded([&] {
    require(ctx != nullptr && id128 != nullptr && out != nullptr, "NULL argument");
    require(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    ctx->use();
    std::unique_ptr<earhip_comm> c(new earhip_comm);
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    EARHIP_NCCL(ncclCommInitRank(&c->comm, world, id, rank));
    EARHIP_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    EARHIP_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    for (auto &e : c->done) EARHIP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &s : c->tev)
      for (auto &e : s) EARHIP_HIP(hipEventCreate(&e));
    *out = c.release();
  });
}

int earhip_comm_destroy(earhip_comm *c) {
  return guarded([&] {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ready) (void)hipEventDestroy(c->ready);
    for (auto e : c->done)
      if (e) (void)hipEventDestroy(e);
    for (auto &s : c->tev)
      for (auto e : s)
        if (e) (void)hipEventDestroy(e);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
  });
}

// rows of the exchange buffers: n_out rounded up to a multiple of the ranks; rank r owns the channels
// [lo, hi) = rows [r * per, (r + 1) * per) clipped to n_out (ragged when the ranks do not divide n_out)
int earhip_comm_channel_range(int n_out, int rank, int world, int *padded_rows, int *lo, int *hi) {
  return guarded([&] {
    require(n_out >= 1 && world >= 1 && rank >= 0 && rank < world, "arguments out of range");
    const int per = (n_out + world - 1) / world;
    if (padded_rows) *padded_rows = per * world;
    if (lo) *lo = std::min(rank * per, n_out);
    if (hi) *hi = std::min((rank + 1) * per, n_out);
  });
}

int earhip_render_exchange_device(earhip_comm *c, int slot, const float *partial_dev, float *owned_dev,
                                  size_t rows_per_rank, size_t row_stride) {
  return guarded([&] {
    require(c != nullptr && partial_dev != nullptr && owned_dev != nullptr, "NULL argument");
    require(slot == 0 || slot == 1, "slot must be 0 or 1");
    require(rows_per_rank >= 1 && row_stride >= 1, "empty exchange");
    c->ctx->use();
    // behind everything enqueued on the context's stream so far (the render that wrote partial_dev) ...
    EARHIP_HIP(hipEventRecord(c->ready, c->ctx->stream));
    EARHIP_HIP(hipStreamWaitEvent(c->xstream, c->ready, 0));
    EARHIP_HIP(hipEventRecord(c->tev[slot][0], c->xstream));
    EARHIP_NCCL(ncclReduceScatter(partial_dev, owned_dev, rows_per_rank * row_stride, ncclFloat, ncclSum, c->comm, c->xstream));
    EARHIP_HIP(hipEventRecord(c->tev[slot][1], c->xstream));
    c->timed[slot] = c->open_[slot] = true;
    // ... and ahead of whatever the caller orders behind earhip_comm_wait(slot)
    EARHIP_HIP(hipEventRecord(c->done[slot], c->xstream));
    c->issued[slot] = true;
  });
}

// The shared loudspeaker bus in one place: after the reduce-scatter rank r holds rows [r * per, (r + 1) * per);
// this collects the owned slices into full_dev [world * per][row_stride] — on every rank (root < 0: one
// ncclAllGather) or on one (root >= 0: the other ranks send their slice straight to it, G - 1 transfers over
// G - 1 different xGMI links at once; full_dev may be NULL on the ranks that are not the root).  Runs on the
// communicator's stream behind the exchange of the same slot and behind everything enqueued on the context's
// stream so far; earhip_comm_wait(slot) orders the context's stream behind it.
int earhip_comm_gather_device(earhip_comm *c, int slot, const float *owned_dev, float *full_dev,
                              size_t rows_per_rank, size_t row_stride, int root) {
  return guarded([&] {
    require(c != nullptr && owned_dev != nullptr, "NULL argument");
    require(slot == 0 || slot == 1, "slot must be 0 or 1");
    require(rows_per_rank >= 1 && row_stride >= 1, "empty gather");
    require(root < c->world, "root out of range");
    require(full_dev != nullptr || (root >= 0 && root != c->rank), "full_dev must not be NULL on a rank that receives");
    c->ctx->use();
    const size_t count = rows_per_rank * row_stride;
    EARHIP_HIP(hipEventRecord(c->ready, c->ctx->stream));
    EARHIP_HIP(hipStreamWaitEvent(c->xstream, c->ready, 0));
    if (!c->open_[slot]) EARHIP_HIP(hipEventRecord(c->tev[slot][0], c->xstream));
    if (root < 0) {
      EARHIP_NCCL(ncclAllGather(owned_dev, full_dev, count, ncclFloat, c->comm, c->xstream));
    } else if (c->rank == root) {
      EARHIP_NCCL(ncclGroupStart());
      ncclResult_t res = ncclSuccess;
      for (int r = 0; r < c->world && res == ncclSuccess; r++)
        if (r != root) res = ncclRecv(full_dev + (size_t)r * count, count, ncclFloat, r, c->comm, c->xstream);
      const ncclResult_t end = ncclGroupEnd();
      EARHIP_NCCL(res);
      EARHIP_NCCL(end);
      if (full_dev + (size_t)root * count != owned_dev)
        EARHIP_HIP(hipMemcpyAsync(full_dev + (size_t)root * count, owned_dev, sizeof(float) * count, hipMemcpyDeviceToDevice,
                                  c->xstream));
    } else {
      EARHIP_NCCL(ncclSend(owned_dev, count, ncclFloat, root, c->comm, c->xstream));
    }
    EARHIP_HIP(hipEventRecord(c->tev[slot][1], c->xstream));
    c->timed[slot] = true;
    c->open_[slot] = false;
    EARHIP_HIP(hipEventRecord(c->done[slot], c->xstream));
    c->issued[slot] = true;
  });
}

// Milliseconds the collectives of `slot` took on the communicator's stream (reduce-scatter, plus the gather
// when one followed): waits for them.  0 when nothing was issued.
int earhip_comm_last_exchange_ms(earhip_comm *c, int slot, double *ms) {
  return guarded([&] {
    require(c != nullptr && ms != nullptr, "NULL argument");
    require(slot == 0 || slot == 1, "slot must be 0 or 1");
    *ms = 0.0;
    if (!c->timed[slot]) return;
    c->ctx->use();
    EARHIP_HIP(hipEventSynchronize(c->tev[slot][1]));
    float t = 0.0f;
    EARHIP_HIP(hipEventElapsedTime(&t, c->tev[slot][0], c->tev[slot][1]));
    *ms = (double)t;
  });
}

// What RCCL itself says about this communicator — for a benchmark line that must show the exchange really ran over N ranks:
// info[0] = ncclCommCount, [1] = ncclCommUserRank, [2] = ncclCommCuDevice (the HIP device it lives on), [3] = ncclGetVersion.
int earhip_comm_info(earhip_comm *c, int info[4]) {
  return guarded([&] {
    require(c != nullptr && info != nullptr, "NULL argument");
    EARHIP_NCCL(ncclCommCount(c->comm, &info[0]));
    EARHIP_NCCL(ncclCommUserRank(c->comm, &info[1]));
    EARHIP_NCCL(ncclCommCuDevice(c->comm, &info[2]));
    EARHIP_NCCL(ncclGetVersion(&info[3]));
  });
}

// The rate ONE link direction gives this communicator, measured: every rank sends `bytes` bytes to rank + shift and receives as
// many from rank - shift (one ncclSend / ncclRecv pair per rank inside a group: G transfers over G different links at once, the
// traffic pattern of the reduce-scatter's and the gather's point-to-point steps), `reps` times between HIP events on the
// communicator's stream after one untimed round.  *GBps = bytes per second a rank SENT (1e9).  Collective: every rank calls
// it with the same arguments.  World size 1: 0.
int earhip_comm_link_probe(earhip_comm *c, size_t bytes, int shift, int reps, double *GBps) {
  return guarded([&] {
    require(c != nullptr && GBps != nullptr, "NULL argument");
    require(bytes >= 4 && bytes <= ((size_t)1 << 32) && reps >= 1 && reps <= 100, "bytes / reps out of range");
    *GBps = 0.0;
    if (c->world < 2) return;
    require(shift % c->world != 0, "shift must not be a multiple of the world size");
    c->ctx->use();
    const size_t count = bytes / 4;
    DevBuf<float> src, dst;
    src.alloc_zero(count, c->xstream);
    dst.alloc_zero(count, c->xstream);
    const int to = ((c->rank + shift) % c->world + c->world) % c->world, from = ((c->rank - shift) % c->world + c->world) % c->world;
    hipEvent_t e[2];
    for (auto &x : e) EARHIP_HIP(hipEventCreate(&x));
    auto round = [&] {
      EARHIP_NCCL(ncclGroupStart());
      const ncclResult_t a = ncclSend(src.p, count, ncclFloat, to, c->comm, c->xstream);
      const ncclResult_t b = ncclRecv(dst.p, count, ncclFloat, from, c->comm, c->xstream);
      const ncclResult_t end = ncclGroupEnd();
      EARHIP_NCCL(a);
      EARHIP_NCCL(b);
      EARHIP_NCCL(end);
    };
    round();
    EARHIP_HIP(hipEventRecord(e[0], c->xstream));
    for (int i = 0; i < reps; i++) round();
    EARHIP_HIP(hipEventRecord(e[1], c->xstream));
    EARHIP_HIP(hipEventSynchronize(e[1]));
    float t = 0.0f;
    EARHIP_HIP(hipEventElapsedTime(&t, e[0], e[1]));
    for (auto &x : e) (void)hipEventDestroy(x);
    if (t > 0.0f) *GBps = (double)bytes * reps / ((double)t * 1e-3) / 1e9;
  });
}

int earhip_comm_wait(earhip_comm *c, int slot) {
  return guarded([&] {
    require(c != nullptr, "comm must not be NULL");
    require(slot == 0 || slot == 1, "slot must be 0 or 1");
    if (!c->issued[slot]) return;
    EARHIP_HIP(hipStreamWaitEvent(c->ctx->stream, c->done[slot], 0));
  });
}

}  // extern "C"
