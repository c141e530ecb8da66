// Probe (not part of the product): semantics of buffer_load_dwordx4 ... lds on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const void* in, unsigned nbytes, unsigned* out, const unsigned* offs) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += blockDim.x) lds[i] = 0xABABABABu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in), 0, nbytes, 0x00020000);
  // wave 0 -> first 1 KiB, wave 1 -> second
  const int wave = threadIdx.x >> 6;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, offs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int N = 4096;
  std::vector<unsigned> h(N); for (int i = 0; i < N; ++i) h[i] = i;
  unsigned *d_in, *d_out, *d_off;
  hipMalloc(&d_in, N * 4); hipMalloc(&d_out, 512 * 4); hipMalloc(&d_off, 128 * 4);
  hipMemcpy(d_in, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> off(128);
  for (int i = 0; i < 128; ++i) off[i] = (unsigned)((i * 37 % 200) * 16);   // scattered 16B chunks
  off[5] = 0x80000000u; off[70] = 0x80000000u; off[9] = N * 4;             // out of range lanes
  hipMemcpy(d_off, off.data(), 128 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d_in, (unsigned)(N * 4), d_out, d_off);
  std::vector<unsigned> o(512);
  hipMemcpy(o.data(), d_out, 512 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 128; ++l) {
    unsigned exp0 = off[l] / 4;
    bool oob = off[l] >= (unsigned)(N * 4);
    unsigned got = o[l * 4];
    if (oob) printf("lane %d OOB: lds = %08x %08x %08x %08x\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    else if (got != exp0 || o[l*4+3] != exp0 + 3) { ++bad; if (bad < 8) printf("lane %d: got %u expected %u\n", l, got, exp0); }
  }
  printf("lane-linear placement mismatches: %d\n", bad);
  return 0;
}
