// Does the duration of ds_bpermute_b32 depend on WHICH lanes are selected?
//
// The fixed-base kernel (device_tables.h, tbl_lds64) and the one-item-per-wavefront kernels (kernels_coop.hip) move the table entry a
// secret digit asks for with ds_bpermute_b32: the secret is the LANE SELECT of the instruction, nothing else.  tools/ct_check.py exempts
// exactly that operand from its "no secret-dependent address" rule on the strength of this measurement: the instruction goes through the
// LDS crossbar without touching LDS memory, so there are no banks to conflict on.  Here: cycles per instruction (s_memtime) for
//   dependent chains (latency: each permute permutes the previous result) and independent streams of 8 (throughput),
//   one wavefront per SIMD and eight,
// under lane-select patterns from the friendliest to the most hostile a bank-conflicting memory would know:
//   identity, reverse, all lanes read lane 0, all read lane 63, l ^ 32 (other half), l ^ 1, l & ~1 (pairs collide), l & 32 (two sources),
//   stride 2 / 4 / 8 / 16 / 32 (mod 64: up to 32 lanes on one source), two random permutations, two random maps with collisions.
//
//   hipcc --offload-arch=gfx950 -O3 -o bpermute_patterns bpermute_patterns.hip && ./bpermute_patterns > profiles/r03/bpermute_patterns.log
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

constexpr int UNROLL = 8;
// dependent chain: v = bpermute(addr, v), UNROLL per iteration
__global__ void __launch_bounds__(1024) k_chain(const int* __restrict__ pattern, int iters, unsigned long long* __restrict__ cycles, unsigned* __restrict__ sink) {
  const int addr = pattern[threadIdx.x & 63] << 2;
  int v = (int)(threadIdx.x * 2654435761u);
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v = __builtin_amdgcn_ds_bpermute(addr, v);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = memtime();
  if (v == 0x12345678) sink[0] = (unsigned)v;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// independent stream: UNROLL permutes of UNROLL different registers per iteration, one wait at the end of the iteration
__global__ void __launch_bounds__(1024) k_stream(const int* __restrict__ pattern, int iters, unsigned long long* __restrict__ cycles, unsigned* __restrict__ sink) {
  const int addr = pattern[threadIdx.x & 63] << 2;
  int v[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) v[u] = (int)(threadIdx.x * 2654435761u) + u;
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(v[u]) : "v"(addr));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = memtime();
  int s = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) s ^= v[u];
  if (s == 0x12345678) sink[0] = (unsigned)s;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  struct Pat { std::string name; std::vector<int> sel; };
  std::vector<Pat> pats;
  auto add = [&](const char* name, auto f) { Pat p{name, std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = f(l) & 63; pats.push_back(p); };
  add("identity", [](int l) { return l; });
  add("reverse", [](int l) { return 63 - l; });
  add("all_read_lane_0", [](int) { return 0; });
  add("all_read_lane_63", [](int) { return 63; });
  add("other_half (l^32)", [](int l) { return l ^ 32; });
  add("neighbour (l^1)", [](int l) { return l ^ 1; });
  add("pairs_collide (l&~1)", [](int l) { return l & ~1; });
  add("two_sources (l&32)", [](int l) { return l & 32; });
  for (int s : {2, 4, 8, 16, 32}) { Pat p{"stride_" + std::to_string(s), std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = (l * s) & 63; pats.push_back(p); }
  srand(12345);
  for (int r = 0; r < 2; ++r) { Pat p{"random_permutation_" + std::to_string(r), std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = l; std::random_shuffle(p.sel.begin(), p.sel.end()); pats.push_back(p); }
  for (int r = 0; r < 2; ++r) { Pat p{"random_map_" + std::to_string(r), std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = rand() & 63; pats.push_back(p); }
  // the patterns the one-item-per-wavefront kernels really issue: coop_table_entry pulls word 10 g' + k from half (idx & 1) — four variants by
  // (idx & 1, negate); the ladder's conditional swap exchanges the halves (l ^ 32 = other_half above) or not (identity)
  for (int v = 0; v < 4; ++v) {
    Pat p{std::string("coop_table_entry idx&1=") + std::to_string(v & 1) + " neg=" + std::to_string(v >> 1), std::vector<int>(64)};
    for (int l = 0; l < 64; ++l) { const int row = l >> 4, k = (l & 15) < 10 ? (l & 15) : 0, g = row < 3 ? row : 0, ge = g < 2 ? g ^ (v >> 1) : g; p.sel[l] = (((v & 1) << 5) | (10 * ge + k)) & 63; }
    pats.push_back(p);
  }
  const size_t n_single = pats.size();
  // families, reported as min / median / max over their members: what the fixed-base kernel issues is a random map (lane l reads entry
  // (neg << 5 | idx) of ITS OWN scalar's digit); "every lane the same digit j" is the batch of one signer
  for (int j = 0; j < 64; ++j) { Pat p{"family:all_read_lane_j", std::vector<int>(64, j)}; pats.push_back(p); }
  for (int r = 0; r < 64; ++r) { Pat p{"family:random_map", std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = rand() & 63; pats.push_back(p); }
  for (int r = 0; r < 32; ++r) { Pat p{"family:random_map_of_32_entries (idx < 32, random sign)", std::vector<int>(64)}; for (int l = 0; l < 64; ++l) p.sel[l] = ((rand() & 1) << 5) | (rand() & 31); pats.push_back(p); }
  for (int r = 0; r < 32; ++r) { Pat p{"family:half_the_lanes_share_a_digit", std::vector<int>(64)}; const int d = rand() & 63; for (int l = 0; l < 64; ++l) p.sel[l] = (l & 1) ? d : (rand() & 63); pats.push_back(p); }
  int* d_pat; unsigned long long* d_cyc; unsigned* d_sink;
  const int max_waves = cus * 4 * 8;
  CK(hipMalloc(&d_pat, 64 * sizeof(int))); CK(hipMalloc(&d_cyc, max_waves * sizeof(unsigned long long))); CK(hipMalloc(&d_sink, 64));
  printf("# %s, %d CUs; cycles of s_memtime per ds_bpermute_b32, median over wavefronts (min..max over wavefronts)\n", prop.name, cus);
  printf("%-26s %30s %30s %30s %30s\n", "lane-select pattern", "chain, 1 wave/SIMD", "chain, 8 waves/SIMD", "stream of 8, 1 wave/SIMD", "stream of 8, 8 waves/SIMD");
  const int iters = 8000;
  std::vector<unsigned long long> host(max_waves);
  std::string fam;
  std::vector<double> fam_med[4];
  auto flush_family = [&] {
    if (fam.empty()) return;
    printf("%-58s", (fam + " (" + std::to_string(fam_med[0].size()) + " patterns: min / median / max of the per-pattern medians)").c_str());
    for (int q = 0; q < 4; ++q) { std::sort(fam_med[q].begin(), fam_med[q].end()); printf("   %7.2f / %7.2f / %7.2f", fam_med[q].front(), fam_med[q][fam_med[q].size() / 2], fam_med[q].back()); fam_med[q].clear(); }
    printf("\n");
    fam.clear();
  };
  for (size_t pi = 0; pi < pats.size(); ++pi) {
    const Pat& p = pats[pi];
    const bool in_family = pi >= n_single;
    if (in_family && p.name != fam) { flush_family(); fam = p.name; }
    CK(hipMemcpy(d_pat, p.sel.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
    if (!in_family) printf("%-26s", p.name.c_str());
    int q = 0;
    for (int kind = 0; kind < 2; ++kind) for (int occ : {1, 8}) {
      const int block = occ == 1 ? 256 : 1024, grid = occ == 1 ? cus : cus * 2;       // 256 threads: one wave per SIMD; 2 x 1024: eight
      const int waves = grid * block / 64;
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(block), 0, 0, d_pat, iters, d_cyc, d_sink);
        else hipLaunchKernelGGL(k_stream, dim3(grid), dim3(block), 0, 0, d_pat, iters, d_cyc, d_sink);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(host.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<double> c(waves);
      for (int w = 0; w < waves; ++w) c[w] = (double)host[w] / ((double)iters * UNROLL);
      std::sort(c.begin(), c.end());
      if (in_family) fam_med[q].push_back(c[waves / 2]);
      else {
        char buf[64];
        snprintf(buf, sizeof(buf), "%.2f (%.2f..%.2f)", c[waves / 2], c.front(), c.back());
        printf(" %30s", buf);
      }
      ++q;
    }
    if (!in_family) printf("\n");
    fflush(stdout);
  }
  flush_family();
  return 0;
}
