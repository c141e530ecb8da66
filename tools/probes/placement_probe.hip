// Probe (not product): which workgroups share a CU when 2 x (512 threads, 72 KB LDS) fit per CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(512, 4) k(unsigned* out, unsigned long long* t) {
  __shared__ unsigned char lds[73728];
  lds[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID (id 4), offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID (id 20)
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  // stay resident for a while so the first wave of blocks fills the chip
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
  if (lds[(threadIdx.x * 7) & 511] == 77) out[0] = 0;
}
int main() {
  const int N = 1024;
  unsigned* d; unsigned long long* dt;
  hipMalloc(&d, N * 8); hipMalloc(&dt, N * 8);
  hipLaunchKernelGGL(k, dim3(N), dim3(512), 0, 0, d, dt);
  std::vector<unsigned> h(2 * N); std::vector<unsigned long long> ht(N);
  hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, N * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  unsigned long long t0 = ht[0]; for (auto v : ht) if (v < t0) t0 = v;
  for (int b = 0; b < N; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu_id;
    cu[key].push_back(b);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) { if (shown++ < 12) { printf("cu %04x:", kv.first); for (int b : kv.second) printf(" %d(t=%llu)", b, (ht[b] - t0)); printf("\n"); } }
  // histogram of the difference between the first two blocks on a CU
  std::map<int, int> hist; for (auto& kv : cu) if (kv.second.size() >= 2) hist[kv.second[1] - kv.second[0]]++;
  printf("block-id distance between first two residents of a CU:"); for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second); printf("\n");
  printf("raw hw[0..3]: %08x %08x %08x %08x xcc %x %x\n", h[0], h[2], h[4], h[6], h[1], h[3]);
  return 0;
}
