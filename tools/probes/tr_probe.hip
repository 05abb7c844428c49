// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS element lands in which lane/slot).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int row_stride_elems) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // 16-lane group g covers a 4x16 block: row = i>>2, col = 16*g + 4*(i&3)
  const short* p = lds + (i >> 2) * row_stride_elems + 16 * g + 4 * (i & 3);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  probe<<<1, 64>>>(d, 64);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // expectation: lane l (i=l&15,g=l>>4), elem j == lds[j*64 + 16*g + i]
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int e = j * 64 + 16 * (l >> 4) + (l & 15); if (h[l*4+j] != e) ++bad; }
  printf("tr_probe mismatches vs expectation: %d\n", bad);
  for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
