// How fast can ONE wave per SIMD issue v_mfma_f32_16x16x4_f32?  (round 6: the tile engine's loop runs ~58 cycles per MFMA at one workgroup per CU)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDS_READS>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ float4 lds[1024];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);
  __syncthreads();
  float4 q[LDS_READS > 0 ? LDS_READS : 1];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (LDS_READS > 0) {
#pragma unroll
      for (int r = 0; r < LDS_READS; ++r) q[r] = lds[(threadIdx.x + 64 * r + it) & 1023];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 32 / NACC; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (LDS_READS > 0) {
#pragma unroll
      for (int r = 0; r < LDS_READS; ++r) a += q[r].x * 1e-30f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int LR>
void run(const char* name, int wgs_per_cu) {
  float* out; unsigned long long* cyc;
  const int grid = 256 * wgs_per_cu, iters = 200;
  hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&cyc, grid * sizeof(unsigned long long));
  hipLaunchKernelGGL((probe<NACC, LR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, LR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = 2.0 * 16 * 16 * 4 * 32.0 * iters * 4 * grid;
  printf("%-34s wgs/cu %d: %.1f cycles per MFMA per wave (wg0), %.1f us, %.1f TFLOP/s\n", name, wgs_per_cu, (double)h[0] / (32.0 * iters), ms * 1e3, flops / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 3; ++w) {
    run<4, 0>("4 accumulators, no LDS", w);
    run<8, 0>("8 accumulators, no LDS", w);
    run<2, 0>("2 accumulators, no LDS", w);
    run<4, 8>("4 acc + 8 ds_read_b128 per 32", w);
  }
  return 0;
}
