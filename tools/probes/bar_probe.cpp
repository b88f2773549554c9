// Probe: can the host write straight into device memory (fine-grained VRAM through the PCIe BAR),
// and how long does a host->GPU->host ping-pong take with (a) pinned host memory both ways,
// (b) VRAM for the host->GPU direction?  Build: hipcc --offload-arch=gfx950 -O2 bar_probe.cpp -o bar_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void pingpong(volatile unsigned* in, volatile unsigned* out, int iters) {
  for (int t = 1; t <= iters; ++t) {
    long spin = 0;
    while (__hip_atomic_load((unsigned*)in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned)t) {
      if (++spin > 50000000L) return;
    }
    __hip_atomic_store((unsigned*)out, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double run(volatile unsigned* in_h, unsigned* in_d, volatile unsigned* out_h, unsigned* out_d, int iters) {
  *in_h = 0; *out_h = 0;
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, in_d, out_d, iters);
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 1; t <= iters; ++t) {
    *in_h = (unsigned)t;
    long spin = 0;
    while (*out_h != (unsigned)t) { if (++spin > 2000000000L) { printf("host timeout at %d\n", t); return -1; } }
  }
  auto t1 = std::chrono::steady_clock::now();
  CK(hipDeviceSynchronize());
  return std::chrono::duration<double>(t1 - t0).count() / iters * 1e6;
}

int main() {
  unsigned *pin_in, *pin_out, *pin_in_d, *pin_out_d;
  CK(hipHostMalloc((void**)&pin_in, 64, hipHostMallocMapped));
  CK(hipHostMalloc((void**)&pin_out, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&pin_in_d, pin_in, 0));
  CK(hipHostGetDevicePointer((void**)&pin_out_d, pin_out, 0));
  printf("pinned<->pinned ping-pong: %.2f us per round trip\n", run(pin_in, pin_in_d, pin_out, pin_out_d, 2000));
  // fine-grained device memory: is it host-writable?
  unsigned* vram = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&vram, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained) -> %s ptr=%p\n", hipGetErrorString(e), (void*)vram);
  if (e != hipSuccess) return 0;
  CK(hipMemset(vram, 0, 4096));
  CK(hipDeviceSynchronize());
  fflush(stdout);
  pid_t pid = fork();
  if (pid == 0) {  // child: try a host store into VRAM; a fault kills only the child
    volatile unsigned* p = vram;
    *p = 12345u;
    unsigned v = *p;
    _exit(v == 12345u ? 0 : 3);
  }
  int status = 0;
  waitpid(pid, &status, 0);
  if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0)) {
    printf("host access to fine-grained VRAM FAULTS (status %d): BAR path not available\n", status);
    return 0;
  }
  printf("host can load/store fine-grained VRAM directly\n");
  printf("VRAM(in) + pinned(out) ping-pong: %.2f us per round trip\n", run(vram, vram, pin_out, pin_out_d, 2000));
  return 0;
}
