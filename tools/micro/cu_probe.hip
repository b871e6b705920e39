// Which workgroups share a CU, and can a workgroup tell that it is the SECOND resident one?  Every workgroup records
// HW_REG_HW_ID, HW_REG_XCC_ID, HW_REG_LDS_ALLOC and its start time; LDS use (79 KB) limits a CU to two workgroups, as the
// convolution kernels do.   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_probe.hip -o tools/micro/cu_probe && ./cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define HWREG(id, off, width) ((id) | ((off) << 6) | (((width) - 1) << 11))

__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
  __shared__ float big[79 * 256];   // 79 KB
  const unsigned hw = __builtin_amdgcn_s_getreg(HWREG(4, 0, 32));
  const unsigned lds = __builtin_amdgcn_s_getreg(HWREG(6, 0, 32));
  const unsigned xcc = __builtin_amdgcn_s_getreg(HWREG(20, 0, 32));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  big[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) acc += big[(threadIdx.x + i) & 255] * 1.0001f;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    unsigned* o = out + blockIdx.x * 8;
    o[0] = hw; o[1] = lds; o[2] = xcc; o[3] = (unsigned)t0; o[4] = (unsigned)(t0 >> 32); o[5] = (unsigned)(t1 - t0);
    o[6] = acc == 12345.f;
  }
}

int main() {
  const int blocks = 2048, spin = 20000;
  unsigned* d;
  hipMalloc(&d, blocks * 8 * 4);
  hipMemset(d, 0, blocks * 8 * 4);
  probe<<<blocks, 256>>>(d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 8);
  hipMemcpy(h.data(), d, blocks * 8 * 4, hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned long long, std::vector<int>> by_cu;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[b * 8], xcc = h[b * 8 + 2] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
  int shown = 0;
  std::map<unsigned, int> lds_vals;
  for (int b = 0; b < blocks; ++b) lds_vals[h[b * 8 + 1]]++;
  printf("distinct LDS_ALLOC values:");
  for (auto& kv : lds_vals) printf(" 0x%08x x%d", kv.first, kv.second);
  printf("\n");
  // first-round residents: the 512 earliest starters; do the two on a CU differ in LDS_ALLOC base?
  int pairs = 0, differ = 0;
  for (auto& kv : by_cu) {
    std::vector<std::pair<unsigned long long, int>> st;
    for (int b : kv.second) st.push_back({((unsigned long long)h[b * 8 + 4] << 32) | h[b * 8 + 3], b});
    std::sort(st.begin(), st.end());
    if (st.size() >= 2) {
      ++pairs;
      const unsigned l0 = h[st[0].second * 8 + 1], l1 = h[st[1].second * 8 + 1];
      differ += (l0 & 0xfff) != (l1 & 0xfff);
      if (shown < 6) {
        printf("cu key %llx: %zu blocks; first two: b%d lds 0x%08x hw 0x%08x t %llu | b%d lds 0x%08x hw 0x%08x t %llu\n", kv.first,
               st.size(), st[0].second, l0, h[st[0].second * 8], st[0].first, st[1].second, l1, h[st[1].second * 8], st[1].first);
        ++shown;
      }
    }
  }
  printf("CUs with >= 2 blocks: %d, first two residents differ in LDS base: %d\n", pairs, differ);
  // block index parity / stride relation of the co-resident pair
  std::map<int, int> delta;
  for (auto& kv : by_cu) {
    std::vector<std::pair<unsigned long long, int>> st;
    for (int b : kv.second) st.push_back({((unsigned long long)h[b * 8 + 4] << 32) | h[b * 8 + 3], b});
    std::sort(st.begin(), st.end());
    if (st.size() >= 2) delta[st[1].second - st[0].second]++;
  }
  printf("blockIdx difference of the first two residents of a CU:");
  for (auto& kv : delta) printf(" %d x%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
