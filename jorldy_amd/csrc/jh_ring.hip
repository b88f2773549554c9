// Asynchronous actor -> learner staging (reference: run_mode.py:212-363 async_distributed_train,
// process.py:7-31,82-97 -- Ray actors -> manager process -> multiprocessing trans_queue -> a `gather_thread` that spins
// on non-atomic flags and re-pickles every transition, then core/agent/ape_x.py:174-199 + PERBuffer.store).
//
// Here: ONE bounded multi-producer / single-consumer ring of transitions in pinned host memory (SoA, the replay
// store's column layout).  Actor threads `jh_ring_produce` rows (+ their actor-side priorities) straight into it --
// lock-free: a ticket per row from one fetch-add, per-slot sequence words publish / recycle the slots (Vyukov's
// bounded queue).  The learner thread `jh_ring_drain`s whatever is published: hipMemcpyAsync from the ring's
// pinned slots directly into the device store's ring (no intermediate copy, no pickling), the sum-tree leaves in the
// same call, and the slots go back to the producers when the copy engine has passed them (an event per drain).
#include <atomic>
#include <chrono>
#include <deque>
#include <thread>

#include "jh_common.h"

struct jh_ring {
  jh_ctx* ctx = nullptr;
  int64_t slots = 0;
  int n_cols = 0;
  int with_priority = 0;
  std::vector<jh_col_desc> cols;
  std::vector<size_t> row_bytes;
  std::vector<char*> host;  // pinned column arrays [slots][row_bytes]
  double* prio = nullptr;   // pinned [slots]
  std::atomic<uint64_t>* seq = nullptr;  // [slots]: == ticket: free for it; == ticket + 1: published
  std::atomic<uint64_t> head{0};         // next ticket to hand out
  uint64_t tail = 0;                     // consumer: next ticket to drain
  struct InFlight {
    hipEvent_t ev;
    uint64_t first;
    int64_t n;
  };
  std::deque<InFlight> inflight;
  std::vector<hipEvent_t> free_events;
  std::atomic<uint64_t> produced{0}, wait_ns{0};
  uint64_t drained = 0;
};

static void ring_release(jh_ring* r, uint64_t first, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t t = first + (uint64_t)i;
    r->seq[t % (uint64_t)r->slots].store(t + (uint64_t)r->slots, std::memory_order_release);
  }
}

// slots whose copies have completed go back to the producers (oldest first: drains complete in stream order)
static void ring_reclaim(jh_ring* r, bool wait_all) {
  while (!r->inflight.empty()) {
    jh_ring::InFlight& f = r->inflight.front();
    if (wait_all) (void)hipEventSynchronize(f.ev);
    else if (hipEventQuery(f.ev) != hipSuccess) { (void)hipGetLastError(); break; }
    ring_release(r, f.first, f.n);
    r->free_events.push_back(f.ev);
    r->inflight.pop_front();
  }
}

JH_EXPORT int jh_ring_create(jh_ctx* ctx, int64_t slots, int32_t n_cols, const jh_col_desc* cols, int32_t with_priority, jh_ring** out) {
  JH_ARG(out && cols);
  JH_ARG(slots > 0 && n_cols > 0 && n_cols <= 16);
  jh_ring* r = new jh_ring();
  r->ctx = ctx;  // may be null: host-only ring (tests, CPU plumbing); pinned allocation needs a device
  r->slots = slots;
  r->n_cols = n_cols;
  r->with_priority = with_priority ? 1 : 0;
  r->cols.assign(cols, cols + n_cols);
  for (int c = 0; c < n_cols; ++c) {
    const size_t rb = jh_dtype_size(cols[c].dtype) * (size_t)cols[c].elems;
    if (rb == 0) {
      delete r;
      return jh_fail(JH_ERR_ARG, "ring column %d has an unknown dtype", c);
    }
    r->row_bytes.push_back(rb);
  }
  for (int c = 0; c < n_cols; ++c) {
    char* p = nullptr;
    const size_t bytes = r->row_bytes[c] * (size_t)slots;
    if (ctx) {
      if (hipHostMalloc((void**)&p, bytes, hipHostMallocDefault) != hipSuccess) p = nullptr;
    } else {
      p = (char*)malloc(bytes);
    }
    if (!p) {
      jh_ring_destroy(r);
      return jh_fail(JH_ERR_NOMEM, "staging ring: %zu bytes of %s host memory for column %d", bytes, ctx ? "pinned" : "pageable", c);
    }
    r->host.push_back(p);
  }
  if (ctx) {
    if (hipHostMalloc((void**)&r->prio, sizeof(double) * (size_t)slots, hipHostMallocDefault) != hipSuccess) r->prio = nullptr;
  } else {
    r->prio = (double*)malloc(sizeof(double) * (size_t)slots);
  }
  if (!r->prio) {
    jh_ring_destroy(r);
    return jh_fail(JH_ERR_NOMEM, "staging ring: priority column");
  }
  r->seq = new std::atomic<uint64_t>[(size_t)slots];
  for (int64_t i = 0; i < slots; ++i) r->seq[i].store((uint64_t)i, std::memory_order_relaxed);
  *out = r;
  return JH_OK;
}

JH_EXPORT void jh_ring_destroy(jh_ring* r) {
  if (!r) return;
  if (r->ctx) {
    (void)hipSetDevice(r->ctx->device);
    ring_reclaim(r, true);
    for (hipEvent_t e : r->free_events) (void)hipEventDestroy(e);
    for (char* p : r->host) (void)hipHostFree(p);
    if (r->prio) (void)hipHostFree(r->prio);
  } else {
    for (char* p : r->host) free(p);
    free(r->prio);
  }
  delete[] r->seq;
  delete r;
}

// Any thread.  Blocks while the ring is full; timeout_ms >= 0 bounds the wait for SPACE (checked before any ticket is
// taken: on JH_ERR_STATE nothing was written); timeout_ms < 0 waits forever.
JH_EXPORT int jh_ring_produce(jh_ring* r, int64_t n, const void* const* h_cols, const double* h_prio, int32_t timeout_ms) {
  JH_ARG(r && h_cols);
  JH_ARG(n >= 0 && n <= r->slots);
  JH_ARG(!r->with_priority || h_prio);
  if (n == 0) return JH_OK;
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t first;
  bool waited = false;
  for (;;) {  // claim n tickets only when their slots are (about to be) free: the oldest unreleased ticket is head - slots
    first = r->head.load(std::memory_order_relaxed);
    const uint64_t last_slot_seq = r->seq[(first + (uint64_t)n - 1) % (uint64_t)r->slots].load(std::memory_order_acquire);
    if (last_slot_seq == first + (uint64_t)n - 1) {
      if (r->head.compare_exchange_weak(first, first + (uint64_t)n, std::memory_order_relaxed)) break;
      continue;
    }
    if (last_slot_seq > first + (uint64_t)n - 1) continue;  // another producer moved head: retry with the new value
    waited = true;
    if (timeout_ms >= 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms))
      return jh_fail(JH_ERR_STATE, "staging ring full for %d ms (%lld slots): the learner is not draining", timeout_ms, (long long)r->slots);
    std::this_thread::yield();
  }
  if (waited) r->wait_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t t = first + (uint64_t)i, slot = t % (uint64_t)r->slots;
    while (r->seq[slot].load(std::memory_order_acquire) != t) std::this_thread::yield();  // earlier slots of the claim: freed in order
    for (int c = 0; c < r->n_cols; ++c) memcpy(r->host[c] + r->row_bytes[c] * slot, (const char*)h_cols[c] + r->row_bytes[c] * (size_t)i, r->row_bytes[c]);
    r->prio[slot] = h_prio ? h_prio[i] : 0.0;
    r->seq[slot].store(t + 1, std::memory_order_release);
  }
  r->produced.fetch_add((uint64_t)n, std::memory_order_relaxed);
  return JH_OK;
}

static int64_t ring_ready(jh_ring* r, int64_t max_rows) {
  int64_t n = 0;
  while (n < max_rows && n < r->slots) {
    const uint64_t t = r->tail + (uint64_t)n;
    if (r->seq[t % (uint64_t)r->slots].load(std::memory_order_acquire) != t + 1) break;
    ++n;
  }
  return n;
}

// The consumer (one thread).  Appends every row published so far (at most max_rows, at most the store's capacity) to
// the device store and, when `per` is given, pushes their leaves (the actors' priorities, or max_priority for a ring
// created without priorities).  Asynchronous on `stream`; the ring slots are recycled once the copies have executed.
JH_EXPORT int jh_ring_drain(jh_ring* r, jh_store* s, jh_per* per, int64_t max_rows, jh_stream stream, int64_t* n_out) {
  JH_ARG(r && s && n_out);
  JH_ARG(r->ctx != nullptr);
  JH_ARG(s->n_cols == r->n_cols);
  for (int c = 0; c < r->n_cols; ++c) JH_ARG(s->row_bytes[c] == r->row_bytes[c]);
  hipStream_t st = jh_s(stream);
  JH_HIP(hipSetDevice(r->ctx->device));
  ring_reclaim(r, false);
  if (max_rows <= 0 || max_rows > s->capacity) max_rows = s->capacity;
  const int64_t n = ring_ready(r, max_rows);
  *n_out = n;
  if (n == 0) return JH_OK;
  int64_t done = 0;
  while (done < n) {  // at most two segments: the staging ring wraps
    const uint64_t slot = (r->tail + (uint64_t)done) % (uint64_t)r->slots;
    int64_t seg = n - done;
    if ((int64_t)slot + seg > r->slots) seg = r->slots - (int64_t)slot;
    std::vector<const void*> src(r->n_cols);
    for (int c = 0; c < r->n_cols; ++c) src[c] = r->host[c] + r->row_bytes[c] * slot;
    int rc = jh_store_append(s, seg, src.data(), hipMemcpyHostToDevice, st);
    if (rc) return rc;
    if (per) {
      rc = jh_per_push(per, seg, r->with_priority ? r->prio + slot : nullptr, stream);
      if (rc) return rc;
    }
    done += seg;
  }
  hipEvent_t ev;
  if (!r->free_events.empty()) {
    ev = r->free_events.back();
    r->free_events.pop_back();
  } else {
    JH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  JH_HIP(hipEventRecord(ev, st));
  r->inflight.push_back(jh_ring::InFlight{ev, r->tail, n});
  r->tail += (uint64_t)n;
  r->drained += (uint64_t)n;
  return JH_OK;
}

// Host-side consumer (tests, CPU plumbing): copies the published rows out and recycles their slots at once.
JH_EXPORT int jh_ring_consume_host(jh_ring* r, int64_t max_rows, void* const* h_out_cols, double* h_prio_out, int64_t* n_out) {
  JH_ARG(r && h_out_cols && n_out);
  if (max_rows <= 0) max_rows = r->slots;
  const int64_t n = ring_ready(r, max_rows);
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t slot = (r->tail + (uint64_t)i) % (uint64_t)r->slots;
    for (int c = 0; c < r->n_cols; ++c) memcpy((char*)h_out_cols[c] + r->row_bytes[c] * (size_t)i, r->host[c] + r->row_bytes[c] * slot, r->row_bytes[c]);
    if (h_prio_out) h_prio_out[i] = r->prio[slot];
  }
  ring_release(r, r->tail, n);
  r->tail += (uint64_t)n;
  r->drained += (uint64_t)n;
  *n_out = n;
  return JH_OK;
}

// Consumer thread: hand the slots of completed drains back to the producers without draining (wait != 0: block
// until every enqueued copy has executed).  jh_ring_drain does the non-blocking form itself on every call.
JH_EXPORT int jh_ring_reclaim(jh_ring* r, int32_t wait) {
  JH_ARG(r != nullptr);
  if (!r->ctx) return JH_OK;
  JH_HIP(hipSetDevice(r->ctx->device));
  ring_reclaim(r, wait != 0);
  return JH_OK;
}

JH_EXPORT int jh_ring_stats(jh_ring* r, int64_t* produced, int64_t* drained, double* producer_wait_ms) {
  JH_ARG(r != nullptr);
  if (produced) *produced = (int64_t)r->produced.load();
  if (drained) *drained = (int64_t)r->drained;
  if (producer_wait_ms) *producer_wait_ms = (double)r->wait_ns.load() * 1e-6;
  return JH_OK;
}
