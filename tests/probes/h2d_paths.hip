// Host -> device copy of a pageable buffer: plain hipMemcpy, hipHostRegister around it, and a pinned double buffer
// filled by memcpy.  (Decides how zada_deflate should bring a large host buffer in.)   hipcc -O2 h2d_paths.hip -o h2d_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 1ull << 30;
  char *h = (char *)malloc(n);
  memset(h, 7, n);
  char *d; hipMalloc(&d, n);
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); double t1 = now();
    printf("pageable hipMemcpy      : %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    t0 = now(); hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double tr = now();
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); double tc = now(); hipHostUnregister(h); t1 = now();
    printf("register+copy+unregister: %.1f ms (register %.1f, copy %.1f, unregister %.1f) rc=%d\n", (t1 - t0) * 1e3, (tr - t0) * 1e3, (tc - tr) * 1e3, (t1 - tc) * 1e3, (int)e);
    const size_t C = 8 << 20; char *p[2]; hipHostMalloc((void **)&p[0], C); hipHostMalloc((void **)&p[1], C);
    hipStream_t s; hipStreamCreate(&s); hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    t0 = now();
    for (size_t o = 0, i = 0; o < n; o += C, i++) {
      const int b = i & 1; if (i >= 2) hipEventSynchronize(ev[b]);
      const size_t c = n - o < C ? n - o : C; memcpy(p[b], h + o, c);
      hipMemcpyAsync(d + o, p[b], c, hipMemcpyHostToDevice, s); hipEventRecord(ev[b], s);
    }
    hipStreamSynchronize(s); t1 = now();
    printf("pinned double buffer    : %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
    hipHostFree(p[0]); hipHostFree(p[1]);
  }
  return 0;
}
