// Probe for VERDICT r3 #2: would ONE persistent cooperative launch per PPO learn() (12 minibatches x {forward, loss, backward,
// norm + Adam}, grid barriers between the phases) beat the 4 launches per minibatch it replaces?  Measures, on this GPU:
//
//   A  launch floor: a dependent chain of trivial kernels (G workgroups x 256 threads) replayed from ONE hipGraph,
//      us per launch -- what a kernel boundary costs here when the kernel body is empty;
//   B  grid barrier inside one launch, G = 16 / 32 / 64 / 128 / 256 resident workgroups, us per barrier (device clock around
//      N barriers, and host wall / N):
//        B1  one monotonic counter, relaxed agent-scope fetch_add + relaxed sc1 poll with s_sleep, NO cache maintenance
//            (a lower bound: nothing a workgroup wrote is visible to another afterwards);
//        B2  the same with the lane-0 release fence before the arrival and the acquire fence after the wait -- the form that
//            actually lets the next phase read what the previous one wrote with plain loads (cdna guide, Guideline 16);
//        B3  XCD-hierarchical: per-XCD arrival counters, the last arriver of an XCD forwards to the top counter, everyone polls
//            one generation word (fewer contenders per word);
//   C  B2 + a phase-sized payload: every workgroup writes 4 KB (its "tile"), barrier, reads ANOTHER workgroup's 4 KB and checks
//      it (correctness of the barrier under test + the cost of the first dependent read after the acquire).
//
// Build: hipcc --offload-arch=gfx950 -O2 grid_barrier_probe.hip -o grid_barrier_probe      Run: ./grid_barrier_probe > table.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e));        \
      exit(2);                                                         \
    }                                                                  \
  } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ void __launch_bounds__(256) k_empty(unsigned* sink) {
  if (sink && threadIdx.x == 1024) sink[0] = 1;  // never true
}

struct Bar {
  unsigned* cnt;      // [0] top counter, [16 * (1 + x)] per-XCD counters, [16 * 9] generation word (64-byte apart)
  unsigned* timeout;  // set when a spin gives up
};

__device__ __forceinline__ bool spin_until_ge(unsigned* p, unsigned target, unsigned* timeout) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load((gu32*)p, RLX_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
    if (spins > 4000000u) {
      __hip_atomic_store((gu32*)timeout, 1u, RLX_AGENT);
      return false;
    }
  }
}

// MODE 1: B1, 2: B2, 3: B3 (with fences)
template <int MODE>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned epoch, int G) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (MODE >= 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 3) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      // how many workgroups of this launch sit on my XCD is not knowable in-kernel: count arrivals per XCD against a census taken
      // in the first barrier (slot [16 * (1 + xcc) + 1] holds it after epoch 1)
      unsigned* xc = b.cnt + 16 * (1 + xcc);
      if (epoch == 1) {
        __hip_atomic_fetch_add((gu32*)(xc + 1), 1u, RLX_AGENT);       // census
        __hip_atomic_fetch_add((gu32*)b.cnt, 1u, RLX_AGENT);          // flat arrival for the first epoch
        ok = spin_until_ge(b.cnt, (unsigned)G, b.timeout);
      } else {
        const unsigned pop = __hip_atomic_load((gu32*)(xc + 1), RLX_AGENT);
        const unsigned t = __hip_atomic_fetch_add((gu32*)xc, 1u, RLX_AGENT);
        if ((t + 1) % pop == 0) __hip_atomic_fetch_add((gu32*)b.cnt, pop, RLX_AGENT);  // last of this XCD forwards its population
        ok = spin_until_ge(b.cnt, epoch * (unsigned)G, b.timeout);
      }
    } else {
      __hip_atomic_fetch_add((gu32*)b.cnt, 1u, RLX_AGENT);
      ok = spin_until_ge(b.cnt, epoch * (unsigned)G, b.timeout);
    }
    if (MODE >= 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_barriers(Bar b, int n_bar, long long* clk) {
  const int G = gridDim.x;
  // warm-up barrier so that every workgroup is resident and past its launch ramp when the clock starts
  if (!grid_barrier<MODE>(b, 1, G)) return;
  const long long t0 = wall_clock64();
  for (int i = 0; i < n_bar; ++i)
    if (!grid_barrier<MODE>(b, 2 + i, G)) return;
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// C: phases with a payload hand-off.  tile [G][1024] floats
__global__ void __launch_bounds__(256) k_phases(Bar b, int n_bar, float* tiles, long long* clk, unsigned* bad) {
  const int G = gridDim.x;
  if (!grid_barrier<2>(b, 1, G)) return;
  const long long t0 = wall_clock64();
  unsigned wrong = 0;
  for (int i = 0; i < n_bar; ++i) {
    float4* mine = reinterpret_cast<float4*>(tiles + (size_t)blockIdx.x * 1024) + threadIdx.x;
    const float v = (float)(i * 1000 + blockIdx.x);
    *mine = make_float4(v, v + 0.25f, v + 0.5f, v + 0.75f);
    if (!grid_barrier<2>(b, 2 + 2 * i, G)) return;
    const int other = (blockIdx.x + 1 + i) % G;
    const float4 q = *(reinterpret_cast<const float4*>(tiles + (size_t)other * 1024) + threadIdx.x);
    const float w = (float)(i * 1000 + other);
    wrong += (q.x != w) || (q.w != w + 0.75f);
    if (!grid_barrier<2>(b, 3 + 2 * i, G)) return;  // nobody overwrites a tile that is still being read
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  if (wrong) atomicAdd(bad, wrong);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned *cnt, *timeout, *bad;
  long long* clk;
  float* tiles;
  CK(hipMalloc(&cnt, 4096));
  CK(hipMalloc(&timeout, 64));
  CK(hipMalloc(&bad, 64));
  CK(hipMalloc(&clk, 8 * 1024));
  CK(hipMalloc(&tiles, sizeof(float) * 1024 * 1024));
  const double clk_mhz = 100.0;  // wall_clock64: constant 100 MHz on gfx9
  const int Gs[5] = {16, 32, 64, 128, 256};
  printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.name, prop.multiProcessorCount);
  bool first = true;
  for (int gi = 0; gi < 5; ++gi) {
    const int G = Gs[gi];
    // ---- A: launch floor (graph of 200 dependent trivial kernels)
    double launch_us = 0;
    {
      hipGraph_t graph;
      hipGraphExec_t exec;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, st, (unsigned*)nullptr);
      CK(hipStreamEndCapture(st, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(exec, st));
      CK(hipStreamSynchronize(st));
      std::vector<double> ts;
      for (int rep = 0; rep < 7; ++rep) {
        const double t0 = now_us();
        CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        ts.push_back((now_us() - t0) / 200.0);
      }
      std::sort(ts.begin(), ts.end());
      launch_us = ts[ts.size() / 2];
      CK(hipGraphExecDestroy(exec));
      CK(hipGraphDestroy(graph));
    }
    // ---- B / C
    const int NB = 400;
    double dev_us[5] = {0, 0, 0, 0, 0}, wall_us[5] = {0, 0, 0, 0, 0};
    unsigned h_bad = 0, h_to = 0;
    for (int mode = 1; mode <= 4; ++mode) {
      std::vector<double> dv, wl;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(cnt, 0, 4096, st));
        CK(hipMemsetAsync(timeout, 0, 64, st));
        CK(hipMemsetAsync(bad, 0, 64, st));
        CK(hipMemsetAsync(clk, 0, 8 * 1024, st));
        CK(hipStreamSynchronize(st));
        Bar b{cnt, timeout};
        const double t0 = now_us();
        if (mode == 1) hipLaunchKernelGGL(k_barriers<1>, dim3(G), dim3(256), 0, st, b, NB, clk);
        if (mode == 2) hipLaunchKernelGGL(k_barriers<2>, dim3(G), dim3(256), 0, st, b, NB, clk);
        if (mode == 3) hipLaunchKernelGGL(k_barriers<3>, dim3(G), dim3(256), 0, st, b, NB, clk);
        if (mode == 4) hipLaunchKernelGGL(k_phases, dim3(G), dim3(256), 0, st, b, NB / 2, tiles, clk, bad);
        CK(hipStreamSynchronize(st));
        const double t1 = now_us();
        std::vector<long long> h(G);
        CK(hipMemcpy(h.data(), clk, sizeof(long long) * G, hipMemcpyDeviceToHost));
        unsigned to = 0, bd = 0;
        CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&bd, bad, 4, hipMemcpyDeviceToHost));
        h_to |= to;
        h_bad += bd;
        long long mx = 0;
        for (int i = 0; i < G; ++i) mx = std::max(mx, h[i]);
        dv.push_back((double)mx / clk_mhz / NB);
        wl.push_back((t1 - t0) / NB);
      }
      std::sort(dv.begin(), dv.end());
      std::sort(wl.begin(), wl.end());
      dev_us[mode] = dv[dv.size() / 2];
      wall_us[mode] = wl[wl.size() / 2];
    }
    printf("%s  {\"workgroups\": %d, \"launch_floor_us_per_kernel_in_graph\": %.3f, \"barrier_no_fence_us\": %.3f, \"barrier_release_acquire_us\": %.3f, "
           "\"barrier_xcd_hier_us\": %.3f, \"barrier_with_4KB_handoff_us\": %.3f, \"host_wall_per_barrier_us\": [%.3f, %.3f, %.3f, %.3f], \"timeouts\": %u, "
           "\"handoff_mismatches\": %u}",
           first ? "" : ",\n", G, launch_us, dev_us[1], dev_us[2], dev_us[3], dev_us[4], wall_us[1], wall_us[2], wall_us[3], wall_us[4], h_to, h_bad);
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
