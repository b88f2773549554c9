// What costs the tile engine's loop its MFMA issue slots at ONE workgroup per CU?  A mimic of the pipelined 64 x 64 x 32 chunk loop (4 waves, 32 MFMAs
// per wave and chunk) with its parts switched on one at a time: barrier | LDS fragment reads | LDS-DMA of the next chunk (16 KB per chunk from an
// L2-resident buffer) | the wait for it.   hipcc --offload-arch=gfx950 -O3 -o gemm_loop_probe gemm_loop_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// BAR: s_barrier per chunk; LDSR: 8 ds_read_b128 per chunk (4 per 16-k step, issued one step ahead); DMA: 4 x 1 KB global_load_lds per wave and chunk;
// NBUF: LDS buffers (DMA prefetch distance NBUF - 1)
template <bool BAR, bool LDSR, bool DMA, int NBUF>
__global__ void __launch_bounds__(256) loop_probe(const float* __restrict__ src, float* out, unsigned long long* cyc, int nchunks) {
  __shared__ __attribute__((aligned(16))) float lds[NBUF * 4096];
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = t; i < NBUF * 4096; i += 256) lds[i] = 1.0f;
  __syncthreads();
  const unsigned lds_base = (unsigned)(uintptr_t)lds + wid * 1024u;
  const float* g = src + ((size_t)blockIdx.x % 64) * 4096 * 64 + (size_t)(wid * 64 + lane) * 4;  // 64 distinct 1 MB streams: L2-resident after the warm-up launch
  float a0[4][4], a1[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a0[i][j] = a1[i][j] = 1.0f + i + j;
  if (DMA) {
    for (int c = 0; c < NBUF && c < nchunks; ++c)
      for (int i = 0; i < 4; ++i) dma16(g + (size_t)c * 4096 + i * 1024, lds_base + c * 16384u + i * 4096u);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * 4) : "memory");
    __syncthreads();
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  int buf = 0;
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const float4* L = reinterpret_cast<const float4*>(lds + buf * 4096);
    // even step
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][0], a0[2 + j][0], acc[i * 2 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (LDSR) for (int i = 0; i < 4; ++i) { const float4 q = L[(lane + 64 * i + 256) & 1023]; a1[i][0] = q.x; a1[i][1] = q.y; a1[i][2] = q.z; a1[i][3] = q.w; }
    __builtin_amdgcn_sched_barrier(0);
    for (int cc = 1; cc < 4; ++cc) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][cc], a0[2 + j][cc], acc[i * 2 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // odd step
    for (int cc = 0; cc < 2; ++cc) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][cc], a1[2 + j][cc], acc[i * 2 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int nb = buf + 1 == NBUF ? 0 : buf + 1;
    if (c + 1 < nchunks) {
      if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * 4) : "memory");
      if (BAR) __syncthreads();
      const float4* Ln = reinterpret_cast<const float4*>(lds + nb * 4096);
      if (LDSR) for (int i = 0; i < 4; ++i) { const float4 q = Ln[(lane + 64 * i) & 1023]; a0[i][0] = q.x; a0[i][1] = q.y; a0[i][2] = q.z; a0[i][3] = q.w; }
      if (DMA && c + NBUF < nchunks)
        for (int i = 0; i < 4; ++i) dma16(g + (size_t)((c + NBUF) & 63) * 4096 + i * 1024, lds_base + buf * 16384u + i * 4096u);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int cc = 2; cc < 4; ++cc) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][cc], a1[2 + j][cc], acc[i * 2 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    buf = nb;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + t] = s;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool BAR, bool LDSR, bool DMA, int NBUF>
void run(const char* name, int wgs_per_cu, const float* src, int nchunks) {
  float* out; unsigned long long* cyc;
  const int grid = 256 * wgs_per_cu;
  (void)hipMalloc(&out, grid * 256 * sizeof(float)); (void)hipMalloc(&cyc, grid * sizeof(unsigned long long));
  hipLaunchKernelGGL((loop_probe<BAR, LDSR, DMA, NBUF>), dim3(grid), dim3(256), 0, 0, src, out, cyc, nchunks);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((loop_probe<BAR, LDSR, DMA, NBUF>), dim3(grid), dim3(256), 0, 0, src, out, cyc, nchunks);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[512]; (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * (grid < 512 ? grid : 512), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-52s wgs/cu %d: %7.0f cycles per chunk (1024 = MFMA-bound at 1 wave/SIMD), kernel %.1f us\n", name, wgs_per_cu, m / nchunks, ms * 1e3);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  float* src; (void)hipMalloc(&src, (size_t)64 * 4096 * 64 * sizeof(float)); (void)hipMemset(src, 0, (size_t)64 * 4096 * 64 * sizeof(float));
  const int nc = 64;
  for (int w = 1; w <= 3; ++w) {
    run<false, false, false, 2>("MFMA only", w, src, nc);
    run<true, false, false, 2>("+ barrier", w, src, nc);
    run<true, true, false, 2>("+ barrier + LDS reads", w, src, nc);
    run<true, true, true, 2>("+ barrier + LDS reads + DMA, 2 buffers", w, src, nc);
    run<true, true, true, 3>("+ barrier + LDS reads + DMA, 3 buffers", w, src, nc);
    if (w < 3) run<true, true, true, 4>("+ barrier + LDS reads + DMA, 4 buffers", w, src, nc);
    run<false, false, true, 3>("MFMA + DMA only, 3 buffers (no barrier: racy)", w, src, nc);
  }
  return 0;
}
