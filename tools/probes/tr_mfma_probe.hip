// GPU-box probe (gfx950): O^T[d][q] = sum_k V[k][d] * P[q][k] for 32 d x 32 queries x 16 keys with ONE v_mfma_f32_32x32x16_bf16 whose A
// operand (V^T fragment) comes from a ROW-MAJOR V[key][d] LDS image through two ds_read_b64_tr_b16 per lane (profiles/r02_tr_b16_probe.md).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_mfma_probe.hip -o tools/probes/tr_mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int S = 160;                 // LDS row pitch in bytes (head_dim 80)
__global__ void probe(const uint16_t* V, const uint16_t* P, float* O) {   // V [16][80] bf16 (d0 = 16: columns 16..47 used), P [32][16] bf16, O [32 d][32 q]
  __shared__ __attribute__((aligned(16))) uint16_t lds[16 * 80];
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 80; i += 64) lds[i] = V[i];
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  const int g = (l >> 4) & 1, half = l >> 5, r = (l & 15) >> 2, c = l & 3, d0 = 16;
  v2u a0, a1;
  const unsigned ad0 = base + (8 * half + 0 + r) * S + (d0 + 16 * g + 4 * c) * 2;
  const unsigned ad1 = base + (8 * half + 4 + r) * S + (d0 + 16 * g + 4 * c) * 2;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1) : "v"(ad0), "v"(ad1) : "memory");
  const uint4 au = make_uint4(a0.x, a0.y, a1.x, a1.y);
  const uint4 bu = *reinterpret_cast<const uint4*>(P + (l & 31) * 16 + half * 8);       // B operand: column q = l & 31, k-slots 8 half .. + 7
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), __builtin_bit_cast(bf16x8, bu), acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) O[((i & 3) + 8 * (i >> 2) + 4 * half) * 32 + (l & 31)] = acc[i];
}
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  uint16_t hv[16 * 80], hp[32 * 16];
  srand(1);
  for (auto& x : hv) x = (uint16_t)(0x3f00 + (rand() & 0xff)) ^ ((rand() & 1) << 15);
  for (auto& x : hp) x = (uint16_t)(0x3f00 + (rand() & 0xff)) ^ ((rand() & 1) << 15);
  uint16_t *dv, *dp; float* dO; float ho[32 * 32];
  (void)hipMalloc(&dv, sizeof(hv)); (void)hipMalloc(&dp, sizeof(hp)); (void)hipMalloc(&dO, sizeof(ho));
  (void)hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice); (void)hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, dp, dO);
  (void)hipMemcpy(ho, dO, sizeof(ho), hipMemcpyDeviceToHost);
  double worst = 0;
  for (int d = 0; d < 32; ++d)
    for (int q = 0; q < 32; ++q) {
      double ref = 0;
      for (int k = 0; k < 16; ++k) ref += (double)bf(hv[k * 80 + 16 + d]) * bf(hp[q * 16 + k]);
      worst = fmax(worst, fabs(ref - ho[d * 32 + q]));
    }
  printf("tr_b16-fed 32x32x16 MFMA vs CPU: max abs err %.3e (%s)\n", worst, worst < 1e-4 ? "MATCH" : "MISMATCH");
  return 0;
}
