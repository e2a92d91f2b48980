// Microbenchmark behind the design of the blocking host call's way back: what the pieces cost on the GPU box.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/host_ubench tools/gpu/host_ubench.hip -lpthread && /tmp/host_ubench
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <functional>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(unsigned nt, size_t n, const std::function<void(size_t, size_t)>& f) {
  if (nt <= 1) { f(0, n); return; }
  std::vector<std::thread> th; const size_t per = (n + nt - 1) / nt;
  for (unsigned k = 1; k < nt; k++) th.emplace_back(f, std::min(n, k * per), std::min(n, (k + 1) * per));
  f(0, std::min(n, per)); for (auto& t : th) t.join();
}
static char* fresh(size_t bytes) { return (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); }
__global__ void spin(long long cycles, double* out) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (out) out[threadIdx.x] = 1.0; }
__global__ void fill(double* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (double)i; }
int main() {
  for (size_t MB : {13, 100}) {
    const size_t bytes = MB << 20, PG = 4096, pages = bytes / PG;
    printf("---- %zu MB (%zu pages)\n", MB, pages);
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) { char* p = fresh(bytes); double t0 = now(); par(nt, pages, [p](size_t a, size_t b) { for (size_t k = a; k < b; k++) ((volatile char*)p)[k * 4096] = 0; }); best = std::min(best, now() - t0); munmap(p, bytes); }
      printf("touch          %2u threads: %.3f ms (%.2f us/page)\n", nt, 1e3 * best, 1e6 * best / pages);
    }
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
      double best = 1e9; int rc = 0;
      for (int rep = 0; rep < 3; rep++) { char* p = fresh(bytes); double t0 = now(); par(nt, pages, [p, &rc](size_t a, size_t b) { if (b > a) rc |= madvise(p + a * 4096, (b - a) * 4096, MADV_POPULATE_WRITE); }); best = std::min(best, now() - t0); munmap(p, bytes); }
      printf("populate_write %2u threads: %.3f ms (rc %d)\n", nt, 1e3 * best, rc);
    }
    { // huge pages hint first
      double best = 1e9; int rc = 0;
      for (int rep = 0; rep < 3; rep++) { char* p = fresh(bytes); rc = madvise(p, bytes, MADV_HUGEPAGE); double t0 = now(); par(4, pages, [p](size_t a, size_t b) { for (size_t k = a; k < b; k++) ((volatile char*)p)[k * 4096] = 0; }); best = std::min(best, now() - t0); munmap(p, bytes); }
      printf("MADV_HUGEPAGE (rc %d) + touch 4 threads: %.3f ms\n", rc, 1e3 * best);
    }
    // pinned -> fresh-but-prefaulted memcpy
    char* pin = nullptr; hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault); memset(pin, 1, bytes);
    for (unsigned nt : {1u, 2u, 4u, 8u}) {
      char* p = fresh(bytes); par(8, pages, [p](size_t a, size_t b) { for (size_t k = a; k < b; k++) ((volatile char*)p)[k * 4096] = 0; });
      double t0 = now(); par(nt, bytes, [p, pin](size_t a, size_t b) { memcpy(p + a, pin + a, b - a); }); double dt = now() - t0;
      printf("memcpy pinned -> prefaulted %u threads: %.3f ms (%.1f GB/s)\n", nt, 1e3 * dt, bytes / dt * 1e-9); munmap(p, bytes);
    }
    { char* p = fresh(bytes); double t0 = now(); par(4, bytes, [p, pin](size_t a, size_t b) { memcpy(p + a, pin + a, b - a); }); double dt = now() - t0;
      printf("memcpy pinned -> UNTOUCHED 4 threads: %.3f ms\n", 1e3 * dt); munmap(p, bytes); }
    // D2H
    double* d = nullptr; hipMalloc((void**)&d, bytes); fill<<<(bytes / 8 + 255) / 256, 256>>>(d, bytes / 8); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) { double t0 = now(); hipMemcpy(pin, d, bytes, hipMemcpyDeviceToHost); double dt = now() - t0; if (rep == 2) printf("D2H into pinned: %.3f ms (%.1f GB/s)\n", 1e3 * dt, bytes / dt * 1e-9); }
    // hipHostRegister of fresh memory, copy straight into it, unregister
    for (int rep = 0; rep < 3; rep++) {
      char* p = fresh(bytes);
      double t0 = now(); hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault); double t1 = now();
      hipError_t e2 = hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t2 = now();
      hipHostUnregister(p); double t3 = now();
      if (rep == 2) printf("hipHostRegister fresh: %.3f ms (%s), D2H into it %.3f ms (%s), unregister %.3f ms\n", 1e3 * (t1 - t0), hipGetErrorName(e), 1e3 * (t2 - t1), hipGetErrorName(e2), 1e3 * (t3 - t2));
      munmap(p, bytes);
    }
    { // 6 separate arrays registered (the caller's arrays are separate allocations)
      const int NA = 6; char* p[NA]; for (int k = 0; k < NA; k++) p[k] = fresh(bytes / NA);
      double t0 = now(); for (int k = 0; k < NA; k++) hipHostRegister(p[k], bytes / NA, hipHostRegisterDefault); double t1 = now();
      for (int k = 0; k < NA; k++) hipHostUnregister(p[k]); double t2 = now();
      printf("register %d arrays of %zu kB: %.3f ms, unregister %.3f ms\n", NA, bytes / NA >> 10, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
      // in parallel threads
      t0 = now(); par(NA, NA, [&](size_t a, size_t b) { for (size_t k = a; k < b; k++) hipHostRegister(p[k], bytes / NA, hipHostRegisterDefault); }); t1 = now();
      par(NA, NA, [&](size_t a, size_t b) { for (size_t k = a; k < b; k++) hipHostUnregister(p[k]); }); t2 = now();
      printf("  the same from %d threads: %.3f ms, unregister %.3f ms\n", NA, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
      for (int k = 0; k < NA; k++) munmap(p[k], bytes / NA);
    }
    { // does a kernel keep running while the host registers / faults?  (kernel of ~2 ms; host work next to it)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      char* p = fresh(bytes);
      hipEventRecord(e0); spin<<<1024, 64>>>(4000000, nullptr); hipEventRecord(e1);
      double t0 = now(); par(8, pages, [p](size_t a, size_t b) { for (size_t k = a; k < b; k++) ((volatile char*)p)[k * 4096] = 0; }); double t1 = now();
      hipEventSynchronize(e1); double t2 = now(); float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("kernel %.3f ms with 8-thread touch beside it (touch %.3f ms, wall to kernel end %.3f ms)\n", ms, 1e3 * (t1 - t0), 1e3 * (t2 - t0)); munmap(p, bytes);
    }
    { // kernel stores straight into mapped pinned memory vs device memory + D2H
      double* hp = nullptr; hipHostMalloc((void**)&hp, bytes, hipHostMallocMapped); double* dp = nullptr; hipHostGetDevicePointer((void**)&dp, hp, 0);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; rep++) { hipEventRecord(e0); fill<<<(bytes / 8 + 255) / 256, 256>>>(dp, bytes / 8); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep == 2) printf("kernel filling mapped host memory: %.3f ms (%.1f GB/s)\n", ms, bytes / ms * 1e-6); }
      hipHostFree(hp);
    }
    hipFree(d); hipHostFree(pin);
  }
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); if (f) { char b[128] = {0}; fgets(b, 127, f); printf("THP enabled: %s", b); fclose(f); }
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
