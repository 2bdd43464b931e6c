// Probe: where does the dispatcher put block b? Prints, per (blockIdx & 7), the histogram of HW_REG_XCC_ID over a launch, for a few
// grid sizes, with and without another kernel occupying part of the chip. (MI355X_MICROARCH.md: "block b runs on XCD b % 8" -- the
// persistent LM kernel checks it per workgroup before it uses XCD-local hand-offs; this probe is the evidence for the default.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ void busy(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 8192 * sizeof(unsigned));
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int busy_blocks : {0, 300}) {
    for (int grid : {8, 79, 474, 632, 768, 3792}) {
      if (busy_blocks) busy<<<busy_blocks, 256, 0, s2>>>(20000ull /* 0.2 ms */, nullptr);
      probe<<<grid, 256, 0, s1>>>(d);
      hipStreamSynchronize(s1);
      hipStreamSynchronize(s2);
      std::vector<unsigned> h(grid);
      hipMemcpy(h.data(), d, grid * sizeof(unsigned), hipMemcpyDeviceToHost);
      int hist[8][16] = {{0}}, mismatch = 0;
      for (int b = 0; b < grid; b++) { hist[b & 7][h[b] & 15]++; mismatch += (h[b] & 15) != (unsigned)(b & 7); }
      printf("grid %4d, %3d busy blocks beside it: %d of %d blocks NOT on XCC_ID == blockIdx %% 8;", grid, busy_blocks, mismatch, grid);
      if (mismatch) for (int r = 0; r < 8; r++) { printf(" [b%%8=%d:", r); for (int x = 0; x < 16; x++) if (hist[r][x]) printf(" xcc%d x%d", x, hist[r][x]); printf("]"); }
      printf("\n");
    }
  }
  return 0;
}
