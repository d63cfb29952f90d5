// TEST INFRASTRUCTURE -- not part of the product library.
//
// A loopback stand-in for the RCCL entry points the C step drivers of libhiprec take as injected function pointers
// (include/hiprec.h: hiprec_nccl_fns = ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd, and the ncclAllReduce
// pointer of hiprec_mf_bpr_dp_epoch_fused_range).  R host threads of ONE process play R ranks on ONE GPU, each with
// its own stream, engine and shard; these functions have RCCL's signatures and RCCL's stream semantics, so the
// N > 1 branches of the drivers (csrc/shard.hip hiprec_shard_planned_steps, csrc/mf.hip the data-parallel epoch)
// execute on the single-GPU test box exactly as they would over xGMI:
//   * send / recv inside a group are collected per thread; group_end publishes the sends (each behind an event of the
//     sender's stream), pairs every recv with the peer's next send in posting order, enqueues "wait for the sender's
//     event, copy device-to-device" on the receiver's stream, and makes the sender's stream wait for the copy (a
//     send buffer may be overwritten by the next kernel of the sender's stream, as after a real ncclSend);
//   * a recv whose element count differs from the matching send, or a peer that never posts, is an ERROR (non-zero
//     return after a time-out), never a hang: a mis-sized plan fails the test;
//   * all_reduce (float32 sum) stages every rank's input, then every rank sums the R staged buffers in rank order --
//     all ranks get bit-identical results, as RCCL guarantees.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

namespace {

constexpr int kMaxWorld = 64;
constexpr int kOk = 0, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5;  // ncclResult_t values

struct SendRec {
  const void* ptr;
  size_t count;
  int dtype;
  hipEvent_t ready;
  hipEvent_t done = nullptr;
  bool consumed = false;
  bool failed = false;
};

struct World {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<SendRec*> q[kMaxWorld][kMaxWorld];  // [src][dst], posting order
  bool failed = false;
  char err[512] = "";
  int timeout_ms = 20000;
  int64_t n_send = 0, n_recv = 0, n_bytes = 0, n_allreduce = 0, n_groups = 0;
  // all-reduce state
  void* staging[kMaxWorld] = {};
  size_t staging_bytes[kMaxWorld] = {};
  hipEvent_t ar_ready[kMaxWorld] = {}, ar_done[kMaxWorld] = {};
  size_t ar_count[kMaxWorld] = {};
  int bar_count = 0;
  uint64_t bar_gen = 0;
};

struct Comm {
  World* w;
  int rank;
};

struct Op {
  bool is_send;
  void* ptr;
  size_t count;
  int dtype, peer;
  Comm* comm;
  hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t dtype_size(int dtype) {
  switch (dtype) {
    case 0: case 1: return 1;          // int8, uint8
    case 2: case 3: case 7: return 4;  // int32, uint32, float32
    case 4: case 5: case 8: return 8;  // int64, uint64, float64
    case 6: case 9: return 2;          // float16, bfloat16
    default: return 0;
  }
}

int fail(World* w, int code, const char* fmt, ...) {
  std::lock_guard<std::mutex> lk(w->mu);
  if (!w->failed) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(w->err, sizeof(w->err), fmt, ap);
    va_end(ap);
    w->failed = true;
  }
  w->cv.notify_all();
  return code;
}

// wait (holding lk) until pred() or the world failed or the time-out passed; true = pred holds
template <class Pred>
bool wait_for(World* w, std::unique_lock<std::mutex>& lk, Pred pred) {
  return w->cv.wait_for(lk, std::chrono::milliseconds(w->timeout_ms), [&] { return w->failed || pred(); }) &&
         !w->failed;
}

bool barrier(World* w) {  // all `world` rank threads; false on time-out / failure
  std::unique_lock<std::mutex> lk(w->mu);
  const uint64_t gen = w->bar_gen;
  if (++w->bar_count == w->world) {
    w->bar_count = 0;
    ++w->bar_gen;
    w->cv.notify_all();
    return !w->failed;
  }
  return wait_for(w, lk, [&] { return w->bar_gen != gen; });
}

int run_group(std::vector<Op>& ops) {
  if (ops.empty()) return kOk;
  World* w = ops[0].comm->w;
  const int me = ops[0].comm->rank;
  std::vector<SendRec*> mine;
  // 1. publish the sends, each behind an event of the sender's stream
  for (Op& op : ops) {
    if (!op.is_send) continue;
    SendRec* r = new SendRec{op.ptr, op.count, op.dtype, nullptr};
    if (hipEventCreateWithFlags(&r->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(r->ready, op.stream) != hipSuccess)
      return fail(w, kInternal, "rank %d: could not record the send event", me);
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->q[me][op.peer].push_back(r);
      ++w->n_send;
    }
    w->cv.notify_all();
    mine.push_back(r);
  }
  // 2. every recv takes the peer's next send in posting order
  for (Op& op : ops) {
    if (op.is_send) continue;
    SendRec* r = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      auto& queue = w->q[op.peer][me];
      if (!wait_for(w, lk, [&] { return !queue.empty(); })) {
        lk.unlock();
        return fail(w, kInternal, "rank %d: no matching send from rank %d for a recv of %zu elements (time-out)", me,
                    op.peer, op.count);
      }
      r = queue.front();
      queue.pop_front();
    }
    if (r->count != op.count || r->dtype != op.dtype) {
      const size_t sent = r->count;
      {
        std::lock_guard<std::mutex> lk(w->mu);
        r->failed = r->consumed = true;
      }
      return fail(w, kInvalidArgument, "rank %d receives %zu elements from rank %d, which sends %zu", me, op.count,
                  op.peer, sent);
    }
    const size_t bytes = op.count * dtype_size(op.dtype);
    hipEvent_t done;
    if (hipStreamWaitEvent(op.stream, r->ready, 0) != hipSuccess ||
        (bytes && hipMemcpyAsync(op.ptr, r->ptr, bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) ||
        hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(done, op.stream) != hipSuccess)
      return fail(w, kInternal, "rank %d: could not enqueue the copy of a recv from rank %d", me, op.peer);
    {
      std::lock_guard<std::mutex> lk(w->mu);
      r->done = done;
      r->consumed = true;
      ++w->n_recv;
      w->n_bytes += static_cast<int64_t>(bytes);
    }
    w->cv.notify_all();
  }
  // 3. the sender's stream continues when its buffers have been read
  size_t k = 0;
  for (Op& op : ops) {
    if (!op.is_send) continue;
    SendRec* r = mine[k++];
    {
      std::unique_lock<std::mutex> lk(w->mu);
      if (!wait_for(w, lk, [&] { return r->consumed; })) {
        lk.unlock();
        return fail(w, kInternal, "rank %d: rank %d never received a send of %zu elements (time-out)", me, op.peer,
                    op.count);
      }
    }
    if (r->failed) return kInvalidArgument;
    if (hipStreamWaitEvent(op.stream, r->done, 0) != hipSuccess)
      return fail(w, kInternal, "rank %d: could not wait for the receiver's copy", me);
    (void)hipEventDestroy(r->ready);
    (void)hipEventDestroy(r->done);
    delete r;
  }
  {
    std::lock_guard<std::mutex> lk(w->mu);
    ++w->n_groups;
  }
  return kOk;
}

constexpr int kSumBlock = 256;
struct Ptrs {
  const float* p[kMaxWorld];
};
__global__ void sum_ranks_kernel(Ptrs src, int world, float* dst, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kSumBlock + threadIdx.x;
  if (i >= n) return;
  float acc = src.p[0][i];
  for (int q = 1; q < world; ++q) acc += src.p[q][i];  // rank order: every rank computes the same bits
  dst[i] = acc;
}

}  // namespace

extern "C" {

void* loopback_world_create(int world) {
  if (world < 1 || world > kMaxWorld) return nullptr;
  World* w = new World;
  w->world = world;
  return w;
}

void loopback_world_destroy(void* world) {
  World* w = static_cast<World*>(world);
  if (!w) return;
  for (int q = 0; q < w->world; ++q) {
    if (w->staging[q]) (void)hipFree(w->staging[q]);
    if (w->ar_ready[q]) (void)hipEventDestroy(w->ar_ready[q]);
    if (w->ar_done[q]) (void)hipEventDestroy(w->ar_done[q]);
  }
  delete w;
}

void* loopback_comm_create(void* world, int rank) {
  World* w = static_cast<World*>(world);
  if (!w || rank < 0 || rank >= w->world) return nullptr;
  return new Comm{w, rank};
}

void loopback_comm_destroy(void* comm) { delete static_cast<Comm*>(comm); }

void loopback_set_timeout_ms(void* world, int ms) { static_cast<World*>(world)->timeout_ms = ms; }

// a rank thread that dies elsewhere calls this so that its peers return instead of waiting for the time-out
void loopback_abort(void* world, const char* why) { fail(static_cast<World*>(world), kInternal, "%s", why ? why : "aborted"); }

int loopback_failed(void* world) { return static_cast<World*>(world)->failed ? 1 : 0; }
const char* loopback_last_error(void* world) { return static_cast<World*>(world)->err; }

// counters: [sends, recvs, bytes copied, all-reduces, completed groups]
void loopback_counters(void* world, int64_t* out) {
  World* w = static_cast<World*>(world);
  std::lock_guard<std::mutex> lk(w->mu);
  out[0] = w->n_send, out[1] = w->n_recv, out[2] = w->n_bytes, out[3] = w->n_allreduce, out[4] = w->n_groups;
}

int loopback_group_start() {
  ++t_depth;
  return kOk;
}

int loopback_group_end() {
  if (t_depth <= 0) return kInvalidUsage;
  if (--t_depth > 0) return kOk;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}

static int post(bool is_send, void* ptr, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return kInvalidArgument;
  if (c->w->failed) return kInternal;
  // (peer == own rank is legal inside a group, as in RCCL: the send is matched by the same group's recv)
  if (peer < 0 || peer >= c->w->world || (peer == c->rank && t_depth <= 0) || dtype_size(dtype) == 0 || (count && !ptr))
    return fail(c->w, kInvalidArgument, "rank %d: bad %s (peer %d, dtype %d, %zu elements)", c->rank,
                is_send ? "send" : "recv", peer, dtype, count);
  t_ops.push_back(Op{is_send, ptr, count, dtype, peer, c, stream});
  if (t_depth > 0) return kOk;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}

int loopback_send(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, comm, stream);
}

int loopback_recv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(false, buf, count, dtype, peer, comm, stream);
}

int loopback_all_reduce(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm,
                        hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return kInvalidArgument;
  World* w = c->w;
  const int me = c->rank, R = w->world;
  if (w->failed) return kInternal;
  if (dtype != 7 || op != 0 || (count && (!sendbuf || !recvbuf)))
    return fail(w, kInvalidArgument, "rank %d: the loopback all-reduce sums float32 only (dtype %d, op %d)", me, dtype, op);
  const size_t bytes = count * sizeof(float);
  // my staging buffer may still be read by the previous round: wait for every rank's previous sum (recorded before
  // the barrier that ended that round)
  for (int q = 0; q < R; ++q)
    if (w->ar_done[q] && hipStreamWaitEvent(stream, w->ar_done[q], 0) != hipSuccess)
      return fail(w, kInternal, "rank %d: all-reduce could not wait for the previous round", me);
  if (w->staging_bytes[me] < bytes) {
    if (w->staging[me]) (void)hipFree(w->staging[me]);  // synchronises the device: nobody reads it any more
    if (hipMalloc(&w->staging[me], bytes) != hipSuccess) return fail(w, kInternal, "rank %d: hipMalloc failed", me);
    w->staging_bytes[me] = bytes;
  }
  if (!w->ar_ready[me] && (hipEventCreateWithFlags(&w->ar_ready[me], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&w->ar_done[me], hipEventDisableTiming) != hipSuccess))
    return fail(w, kInternal, "rank %d: hipEventCreate failed", me);
  if ((bytes && hipMemcpyAsync(w->staging[me], sendbuf, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
      hipEventRecord(w->ar_ready[me], stream) != hipSuccess)
    return fail(w, kInternal, "rank %d: could not stage the all-reduce input", me);
  w->ar_count[me] = count;
  if (!barrier(w)) return fail(w, kInternal, "rank %d: a peer never reached the all-reduce of %zu elements (time-out)", me, count);
  Ptrs src;
  for (int q = 0; q < R; ++q) {
    if (w->ar_count[q] != count)
      return fail(w, kInvalidArgument, "rank %d reduces %zu elements, rank %d reduces %zu", me, count, q, w->ar_count[q]);
    src.p[q] = static_cast<const float*>(w->staging[q]);
    if (hipStreamWaitEvent(stream, w->ar_ready[q], 0) != hipSuccess)
      return fail(w, kInternal, "rank %d: could not wait for rank %d's input", me, q);
  }
  if (count) {
    const unsigned grid = static_cast<unsigned>((count + kSumBlock - 1) / kSumBlock);
    sum_ranks_kernel<<<grid, kSumBlock, 0, stream>>>(src, R, static_cast<float*>(recvbuf), count);
    if (hipGetLastError() != hipSuccess) return fail(w, kInternal, "rank %d: the sum kernel did not launch", me);
  }
  if (hipEventRecord(w->ar_done[me], stream) != hipSuccess) return fail(w, kInternal, "rank %d: event record failed", me);
  if (me == 0) {
    std::lock_guard<std::mutex> lk(w->mu);
    ++w->n_allreduce;
  }
  if (!barrier(w)) return fail(w, kInternal, "rank %d: a peer never finished the all-reduce (time-out)", me);
  return kOk;
}

}  // extern "C"
