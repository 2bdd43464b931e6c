// Probe: how many cycles does one SIMD need per wave64 VALU instruction on gfx950? (The VALU roofline of the k-NN / RBF kernels needs the
// peak issue rate: 2 cycles per instruction -> 78.6 T lane-op/s, 4 cycles -> 39.3.) Independent v_fma_f32 chains, W waves per SIMD,
// cycles from s_memtime inside the kernel and from HIP events outside. Also the same for v_pk_fma_f32 and for a 64-bit compare + cndmask pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHAINS 16
template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters) {
  float a[CHAINS];
  for (int i = 0; i < CHAINS; i++) a[i] = threadIdx.x * 0.001f + i;
  const float m = 1.0001f, c = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < CHAINS; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    } else {
#pragma unroll
      for (int i = 0; i < CHAINS; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {a[i], a[i + 1]}, mm = {m, m}, cc = {c, c};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
        a[i] = v.x; a[i + 1] = v.y;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CHAINS; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int iters = 4096;
  float* d; unsigned long long* c;
  hipMalloc(&d, 256 * 16 * 256 * sizeof(float)); hipMalloc(&c, 256 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; kind++)
    for (int wg_per_cu : {1, 2, 4, 8}) {   // 256-thread workgroups = 4 waves = one wave per SIMD each
      const int grid = 256 * wg_per_cu;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (kind == 0) rate<0><<<grid, 256>>>(d, c, iters); else rate<1><<<grid, 256>>>(d, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(grid);
      hipMemcpy(h.data(), c, grid * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= grid;
      const double insts_per_wave = (double)iters * (kind == 0 ? CHAINS : CHAINS / 2);
      // s_memtime ticks at a constant 100 MHz on gfx9 (REFCLK): report both ticks and the event-based figure
      printf("%s  %d waves/SIMD: %.3f ms, %.2f G wave-instr/s chip = %.3f instr per SIMD per ns; s_memtime ticks per wave %.0f (%.4f ticks per instr per resident wave)\n",
             kind == 0 ? "v_fma_f32   " : "v_pk_fma_f32", wg_per_cu, ms, insts_per_wave * grid * 4 / (ms * 1e-3) / 1e9, insts_per_wave * grid * 4 / 1024.0 / (ms * 1e6), avg, avg / insts_per_wave / wg_per_cu);
    }
  printf("reading: instr per SIMD per ns x (cycles per instr) = clock in GHz; at 2.4 GHz, 1.2 -> 2 cycles per wave64 instruction, 0.6 -> 4 cycles\n");
  return 0;
}
