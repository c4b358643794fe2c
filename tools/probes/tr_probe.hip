// GPU-box probe (gfx950): which LDS elements does ds_read_b64_tr_b16 hand to each lane?  LDS is filled with bf16 slots whose bits are their
// own index; every lane supplies a byte address per pattern and prints the four 16-bit indices it receives.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o tools/probes/tr_probe.bin ; run on the box: tools/probes/tr_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out, int pattern) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = l * 8;                                                     // 16 lanes cover 128 contiguous bytes (4 x 16 row-major)
  else if (pattern == 1) addr = ((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 256;  // rows of a group 64 B apart
  else if (pattern == 2) addr = ((l & 15) >> 2) * 32 + (l & 3) * 8 + (l >> 4) * 1024; // [key][16 d] image, 32-B rows, groups = d slabs
  else addr = (l & 15) * 32 + (l >> 4) * 8;                                           // one row per lane, 32-B rows
  addr += (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;      // LDS byte offset of the array
  v2u r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (lane: 4 element indices)\n", p);
    for (int l = 0; l < 64; ++l) printf("%2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
  }
  return 0;
}
