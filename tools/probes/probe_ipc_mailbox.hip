// Feasibility probe for the peer-mapped mailbox: two PROCESSES on one node (same GPU or two GPUs), each owns a
// fine-grained device buffer exported with hipIpcGetMemHandle; each runs ONE kernel that ping-pongs a tagged 32-double row with
// the peer for N rounds (system-scope stores into the peer's buffer, system-scope polling of its own) -- i.e. what the opener
// workgroup of the persistent LM kernel does once per trip. Prints the round-trip time, or the watchdog abort.
//   hipcc --offload-arch=gfx950 -O2 -o probe_ipc_mailbox probe_ipc_mailbox.hip && ./probe_ipc_mailbox [dev0 dev1] [rounds]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", getpid(), #x, hipGetErrorString(e_)); _exit(3); } } while (0)

__global__ void pingpong(double* mine, double* peer, int rank, int rounds, unsigned long long watchdog, unsigned long long* out) {
  const int t = threadIdx.x;  // 64 threads
  unsigned long long t0 = wall_clock64();
  int ok = 1;
  for (int r = 1; r <= rounds && ok; r++) {
    // write my row (32 values + tag) into the peer's slot `rank`
    if (t < 32) __hip_atomic_store(&peer[rank * 40 + t], (double)(r * 1000 + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (t == 0) __hip_atomic_store(&peer[rank * 40 + 32], (double)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for the peer's row of this round in MY buffer
    if (t == 0) {
      const unsigned long long w0 = wall_clock64();
      while (__hip_atomic_load(&mine[(1 - rank) * 40 + 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (double)r) {
        if (wall_clock64() - w0 > watchdog) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    ok = __shfl(ok, 0);
    if (ok && t < 32) {
      const double v = __hip_atomic_load(&mine[(1 - rank) * 40 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v < (double)(r * 1000 + t)) ok = 0;  // payload older than its tag
    }
    ok = __all(ok);
  }
  if (t == 0) { out[0] = wall_clock64() - t0; out[1] = ok; }
}

int main(int argc, char** argv) {
  const int dev[2] = {argc > 2 ? atoi(argv[1]) : 0, argc > 2 ? atoi(argv[2]) : 0};
  const int rounds = argc > 3 ? atoi(argv[3]) : (argc == 2 ? atoi(argv[1]) : 200);
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 2;
  const pid_t pid = fork();
  const int rank = pid == 0 ? 1 : 0;
  const int rd = rank ? p2c[0] : c2p[0], wr = rank ? c2p[1] : p2c[1];
  CK(hipSetDevice(dev[rank]));
  double* mine = nullptr;
  CK(hipExtMallocWithFlags((void**)&mine, 4096, hipDeviceMallocFinegrained));
  CK(hipMemset(mine, 0, 4096));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t hm, hp;
  CK(hipIpcGetMemHandle(&hm, mine));
  if (write(wr, &hm, sizeof(hm)) != (ssize_t)sizeof(hm) || read(rd, &hp, sizeof(hp)) != (ssize_t)sizeof(hp)) return 2;
  double* peer = nullptr;
  CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
  unsigned long long* out = nullptr;
  CK(hipHostMalloc((void**)&out, 64, hipHostMallocDefault));
  out[0] = out[1] = 0;
  char go = 1;  // both sides mapped: start together
  if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 2;
  pingpong<<<1, 64>>>(mine, peer, rank, rounds, 20'000'000ull /* 200 ms */, out);
  CK(hipDeviceSynchronize());
  printf("[rank %d, device %d] %s: %d rounds in %.1f us -> %.2f us per exchange\n", rank, dev[rank], out[1] ? "ok" : "WATCHDOG/STALE", rounds, out[0] / 100.0, out[0] / 100.0 / rounds);
  fflush(stdout);
  if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 2;  // peer done with my buffer
  CK(hipIpcCloseMemHandle(peer));
  CK(hipFree(mine));
  if (pid != 0) { int st = 0; waitpid(pid, &st, 0); return WEXITSTATUS(st); }
  return out[1] ? 0 : 1;
}
