// Probe (not a product file): do LDS atomics of one wave instruction that hit the same address return their
// pre-op values in lane order?  (hipcc --offload-arch=gfx950 lds_atomic_order.hip -o /tmp/lds_atomic_order && /tmp/lds_atomic_order)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t seed, uint32_t *bad, int ndig) {
  __shared__ uint32_t cnt[256];
  const int lane = threadIdx.x & 63;
  uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + threadIdx.x * 9973u;
  for (int it = 0; it < 2000; it++) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t d = x % (uint32_t)ndig;
    // only wave 0 of the block takes part (one instruction at a time)
    if (threadIdx.x < 64) {
      const uint32_t r = atomicAdd(&cnt[d], 1u);
      // expected rank = number of lower lanes with the same digit
      uint32_t exp = 0;
      for (int l = 0; l < 64; l++) { const uint32_t dl = __shfl(d, l); if (l < lane && dl == d) exp++; }
      if (r != exp) atomicAdd(bad, 1u);
    }
    __syncthreads();
  }
}
int main() {
  uint32_t *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  for (int nd : {1, 2, 3, 7, 16, 64, 256}) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, 12345u + nd, bad, nd);
    uint32_t h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("digits %3d: out-of-lane-order results %u of %d\n", nd, h, 512 * 2000 * 64);
  }
  return 0;
}
