// checks the lane mapping of ds_read_b64_tr_b16 as described in cdna_hip_programming.md T10:
// per 16-lane group, lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of rows 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;   // value = row*64 + col
  __syncthreads();
  const int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4;
  // group g reads the 4x16 block at rows 8g..8g+3, columns 16..31
  const int row = 8 * g + (i16 >> 2), col = 16 + 4 * (i16 & 3);
  unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (row * 64 + col) * 2;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = v.x & 0xFFFF; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xFFFF; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
  unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int g = lane >> 4, i = lane & 15;
    for (int j = 0; j < 4; ++j) { int want = (8 * g + j) * 64 + 16 + i; if (h[lane * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d got (r%d,c%d) want (r%d,c%d)\n", lane, j, h[lane*4+j] / 64, h[lane*4+j] % 64, want / 64, want % 64); ++bad; } }
  }
  printf("tr read mapping: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as documented", bad);
  return 0;
}
