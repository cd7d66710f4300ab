// ds_read_b64_tr_b8 (gfx950) semantics probe for the fp8 weight gradient: lds[byte e] = e & 0xff with a known layout, every lane
// issues one transposing read at a chosen address, prints what it received.
//   variant 1: each 16-lane group g reads an 8-row x 16-byte block, row stride RS: addr = g*BLK + (q>>1)*RS + (q&1)*8
//   variant 2: transposed assignment:                                             addr = g*BLK + (q&7)*RS + (q>>3)*8
// build: hipcc --offload-arch=gfx950 -O2 tools/hwprobe/tr8probe.hip -o tools/hwprobe/tr8probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) i2v lds_i2;

__global__ void k_tr8(unsigned char* out, unsigned short* src_idx, int variant, int RS, int BLK) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];      // value = element index (16-bit view to identify bytes)
  // byte e holds (e % 251) so that neighbouring bytes differ; we also export the byte's index through a second pass
  unsigned char* b = reinterpret_cast<unsigned char*>(lds);
  for (int i = threadIdx.x; i < 32768; i += 64) b[i] = (unsigned char)(i % 251);
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, q = l & 15;
  int addr = (variant == 1) ? g * BLK + (q >> 1) * RS + (q & 1) * 8 : g * BLK + (q & 7) * RS + (q >> 3) * 8;
  i2v v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)((__attribute__((address_space(3))) char*)lds + addr));
  const unsigned char* vb = reinterpret_cast<const unsigned char*>(&v);
  for (int j = 0; j < 8; j++) out[l * 8 + j] = vb[j];
  (void)src_idx;
}

int main() {
  unsigned char* d; hipMalloc(&d, 64 * 8);
  for (int variant = 1; variant <= 2; ++variant) {
    const int RS = 64, BLK = 1024;
    k_tr8<<<1, 64>>>(d, nullptr, variant, RS, BLK);
    std::vector<unsigned char> h(512);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("variant %d (RS %d, BLK %d): lane -> the 8 bytes it received, decoded as (row, col) of its group's 8x16 block\n", variant, RS, BLK);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 8; ++j) {
        // find the byte index within the group's block whose value matches (values are unique within 251 bytes; block rows are 64 apart)
        int g = l >> 4, found = -1;
        for (int r = 0; r < 8 && found < 0; ++r)
          for (int c = 0; c < 16; ++c)
            if ((unsigned char)((g * BLK + r * RS + c) % 251) == h[l * 8 + j]) { found = r * 16 + c; break; }
        printf(" (%d,%2d)", found >> 4, found & 15);
      }
      printf("\n");
    }
  }
  return 0;
}
