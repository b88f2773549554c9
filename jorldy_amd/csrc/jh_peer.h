// Device-side pieces of the peer-pointer collectives (jh_peer.hip) that other kernels of the library fold into their own launches:
// the argument block, the bounded flag wait, and the <= 16-float mailbox exchange (jh_ppo.hip's critic select does it in its prologue).
#pragma once
#include "jh_common.h"

constexpr int kPeerMaxRanks = 16;
constexpr int kPeerSmallMax = 16;
constexpr int kPeerCtlBytes = 16384;
// control block at the head of every arena (offsets in bytes); written by PEERS, polled by the owner
constexpr int kPeerOffFlagsIn = 0, kPeerOffFlagsOut = 256, kPeerOffFlagsSmall = 512, kPeerOffSmallBox = 1024;  // small_box [2][kPeerMaxRanks][kPeerSmallMax] floats = 2 KB

struct PeerArgs {
  char* arena[kPeerMaxRanks];  // every rank's arena as THIS process sees it (own: the allocation itself)
  int nranks, rank;
  int64_t n, slice;        // floats in the bucket, floats per slice (multiple of 4)
  size_t off_in, off_out, out_stride;  // byte offsets of `in` and `out[2]` in an arena
  unsigned* seq;       // [4] device-private: [0] completed all-reduces, [1] completed small exchanges
  unsigned* arrive;    // [4] device-private arrival counters
  unsigned* err;       // [1] bounded waits that gave up
  float* bucket;
};

__device__ __forceinline__ unsigned jh_ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void jh_st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// thread 0 of the workgroup waits until words[p] >= seq for every rank p (its own included); everybody leaves with an acquire
__device__ __forceinline__ void jh_peer_wait_all(const unsigned* words, int nranks, unsigned seq, unsigned* err) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < nranks; ++p) {
      while ((int)(jh_ld_sys(words + p) - seq) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s of the 100 MHz clock
          atomicAdd(err, 1u);
          p = nranks;
          break;
        }
      }
    }
  }
  __syncthreads();
  // (no acquire fence: everything a peer published is read with system-scope loads that bypass this GPU's caches -- a system-scope fence would write
  // back / invalidate the whole L2 around every hand-off, which costs more than the exchange: the lesson of jh_tgemm.hip's split-K hand-off)
}
// 8-byte system-scope (write-through / cache-bypassing) accesses: what crosses between GPUs
__device__ __forceinline__ void jh_st8_sys(void* p, unsigned long long v) { __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long jh_ld8_sys(const void* p) { return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void jh_st_f2_sys(float* p, float a, float b) { jh_st8_sys(p, (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32)); }
__device__ __forceinline__ void jh_ld_f2_sys(const float* p, float& a, float& b) {
  const unsigned long long v = jh_ld8_sys(p);
  a = __uint_as_float((unsigned)v);
  b = __uint_as_float((unsigned)(v >> 32));
}


// The <= 16-float exchange as a device function for a SINGLE workgroup (any block size >= n): vals[t] <- sum (x scale) over the ranks, in rank
// order, of the ranks' vals[t]; every thread of the workgroup must call it (it holds barriers).  Same mailboxes / sequence counter as
// jh_peer_small_kernel: the ranks must run the same sequence of small exchanges.  nranks == 1: nothing is sent.
__device__ __forceinline__ float jh_peer_small_exchange(const PeerArgs& a, float mine, int n, float scale) {
  const int t = threadIdx.x;
  if (a.nranks <= 1) return mine * scale;
  const unsigned seq = a.seq[1] + 1u;
  const size_t box = kPeerOffSmallBox + (size_t)(seq & 1u) * kPeerMaxRanks * kPeerSmallMax * sizeof(float);
  if (t < n)
    for (int p = 0; p < a.nranks; ++p)
      __hip_atomic_store((float*)(a.arena[p] + box) + a.rank * kPeerSmallMax + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have completed
  __syncthreads();
  if (t == 0)
    for (int p = 0; p < a.nranks; ++p) jh_st_sys((unsigned*)(a.arena[p] + kPeerOffFlagsSmall) + a.rank, seq);
  jh_peer_wait_all((const unsigned*)(a.arena[a.rank] + kPeerOffFlagsSmall), a.nranks, seq, a.err);
  float s = 0.f;
  if (t < n)
    for (int p = 0; p < a.nranks; ++p)
      s += __hip_atomic_load((const float*)(a.arena[a.rank] + box) + p * kPeerSmallMax + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (t == 0) a.seq[1] = seq;
  return s * scale;
}

struct jh_peer;
// host side (jh_peer.hip): the argument block of a connected communicator for a kernel of another translation unit
int jh_peer_args_for(jh_peer* p, PeerArgs* out);
