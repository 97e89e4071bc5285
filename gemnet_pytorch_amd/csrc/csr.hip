// CSR groupings of the index plans (gemnet_pytorch_amd/graph.py) built natively (include/gemnet_hip.h, gn_csr_*).
//
// Every gather of the path (`x[id3_expand_ba]`, `x[id4_expand_abd]`, h[id_a] ...: interaction_block.py:543,548,562,678,693,
// embedding_block.py:70-71) needs the transposed grouping — permutation + offsets by destination row — for its adjoint
// (no atomics).  The plan of a batch holds ~10 of them, and the dynamic-shape paths rebuild the plan for
// every batch INSIDE the replayed hipGraph (padded.py): as torch ops each grouping was a full 32/64-bit `argsort` + a gather of
// the sorted keys + an `arange` + a `searchsorted` + two dtype conversions (~13 launches; ~60 per replay of GemNet-T, a 9 M-key
// 64-bit-index sort per replay of GemNet-Q).  Here: ONE rocPRIM radix sort of (key, position) pairs over the SIGNIFICANT bits
// of the keys only (ceil(log2(n_rows)): 15 bits for 18 k edges, 20 for 0.6 M intermediate triplets — not 32), stable, int32
// throughout, + one lower-bound kernel for the offsets.  Caller-owned workspace, no allocation, no synchronisation: capturable.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace {

// seg_off[r] = number of keys < r  (keys sorted ascending), r = 0 .. n_rows
__global__ __launch_bounds__(256) void lower_bound_kernel(const int32_t* __restrict__ sorted, int64_t n, int64_t n_rows,
                                                         int32_t* __restrict__ seg_off) {
  for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * 256) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    seg_off[r] = (int32_t)lo;
  }
}

inline unsigned key_bits(int64_t n_rows) {
  unsigned b = 1;
  while (b < 31 && ((int64_t)1 << b) < n_rows) ++b;
  return b;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// rocPRIM's default switches to a MERGE sort below 2^20 items: ~21 launches of 5-6 us.  Measured (profiles/r5_csr_onesweep_ab.txt):
// for the 100 k .. 600 k keys of a batch's triplet / intermediate-triplet groupings Onesweep over the <= 20 significant bits (a
// histogram + two or three passes) is faster (GemNet-Q new-batch replay 16.30 -> 16.07 ms), for the few thousand keys of a single
// molecule's plan it is slower (MD step 2.40 -> 2.56 ms): the choice is made per call by size.
//   GN_CSR_ONESWEEP_FROM: smallest item count sorted by Onesweep (A/B builds: 1048577 = rocPRIM's default everywhere).
#ifndef GN_CSR_ONESWEEP_FROM
#define GN_CSR_ONESWEEP_FROM 100000
#endif
using cfg_onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)4096>;
using cfg_default = rocprim::default_config;

template <class Cfg>
inline hipError_t sort_pairs(void* temp, size_t& bytes, const int32_t* keys, int32_t* sorted, int32_t* perm, int64_t n, unsigned bits,
                             hipStream_t st) {
  return rocprim::radix_sort_pairs<Cfg>(temp, bytes, keys, sorted, rocprim::counting_iterator<int32_t>(0), perm, (size_t)n, 0u, bits, st);
}
inline bool use_onesweep(int64_t n) { return n >= (int64_t)GN_CSR_ONESWEEP_FROM; }

inline size_t sort_temp_bytes(int64_t n, unsigned bits) {
  size_t bytes = 0;
  if (use_onesweep(n)) (void)sort_pairs<cfg_onesweep>(nullptr, bytes, nullptr, nullptr, nullptr, n, bits, nullptr);
  else (void)sort_pairs<cfg_default>(nullptr, bytes, nullptr, nullptr, nullptr, n, bits, nullptr);
  return bytes;
}

// ---- CSR of items grouped by edge, by the row of their edge, without sorting the item keys -----------------------------------
// The triplets are sorted by their reduce edge (item range of edge e: so[e] .. so[e+1]); their CSR by the ATOM of that edge is
// the edge CSR (perm_e, seg_e: E keys) expanded into item ranges: slot i of the edge CSR owns the items so[order[i]] .. and
// lands at off[i] = sum of the item counts of the slots before it — the permutation of the stable sort of the T item keys.
// (graph.expanded_csr did this with ~13 ATen launches per grouping, two groupings per plan, inside every dynamic replay.)
__global__ __launch_bounds__(1024) void expanded_scan_kernel(const int32_t* __restrict__ perm_e, const int32_t* __restrict__ so,
                                                             int64_t E, int32_t* __restrict__ off) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < E; base += 1024) {
    const int64_t i = base + tid;
    int64_t v = 0;
    if (i < E) {
      const int64_t e = perm_e ? perm_e[i] : i;
      v = (int64_t)so[e + 1] - so[e];
    }
    int64_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(s, o, 64);
      if (lane >= o) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t carry = carry_s;
    if (i < E) off[i] = (int32_t)(carry + woff + s - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + s;
    __syncthreads();
  }
  if (tid == 0) off[E] = (int32_t)carry_s;
}

// 16 lanes per slot of the edge CSR write its item range; the first n_rows + 1 threads also write the row offsets
__global__ __launch_bounds__(256) void expanded_fill_kernel(const int32_t* __restrict__ perm_e, const int32_t* __restrict__ seg_e,
                                                            const int32_t* __restrict__ so, const int32_t* __restrict__ off,
                                                            int64_t E, int64_t n_rows, int32_t* __restrict__ perm,
                                                            int32_t* __restrict__ seg_out) {
  const int64_t gid = blockIdx.x * (int64_t)256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  for (int64_t r = gid; r <= n_rows; r += nth) seg_out[r] = off[seg_e[r]];
  const int sub = threadIdx.x & 15;
  for (int64_t i = gid >> 4; i < E; i += nth >> 4) {
    const int64_t e = perm_e ? perm_e[i] : i;
    const int32_t first = so[e], cnt = so[e + 1] - first, o = off[i];
    for (int32_t j = sub; j < cnt; j += 16) perm[o + j] = first + j;
  }
}

}  // namespace

extern "C" int gn_expanded_csr_i32(const int32_t* perm_e, const int32_t* seg_e, int64_t n_rows, const int32_t* seg_off_of_edge,
                                   int64_t E, int32_t* perm, int32_t* seg_out, int32_t* ws, void* stream) {
  if (E < 0 || n_rows < 0 || !ws) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(expanded_scan_kernel, dim3(1), dim3(1024), 0, st, perm_e, seg_off_of_edge, E, ws);
  int64_t nb = (E * 16 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(expanded_fill_kernel, dim3((unsigned)nb), dim3(256), 0, st, perm_e, seg_e, seg_off_of_edge, (const int32_t*)ws, E,
                     n_rows, perm, seg_out);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t gn_csr_ws_bytes(int64_t n, int64_t n_rows) {
  if (n <= 0) return 256;
  return (int64_t)(align256((size_t)n * sizeof(int32_t)) + align256(sort_temp_bytes(n, key_bits(n_rows))) + 256);
}

extern "C" int gn_seg_offsets_i32(const int32_t* sorted_keys, int64_t n, int64_t n_rows, int32_t* seg_off, void* stream) {
  if (n_rows < 0 || n < 0) return (int)hipErrorInvalidValue;
  int64_t nb = (n_rows + 1 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(lower_bound_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), sorted_keys, n,
                     n_rows, seg_off);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_csr_build_i32(const int32_t* keys, int64_t n, int64_t n_rows, int32_t* perm, int32_t* seg_off, void* ws,
                                int64_t ws_bytes, void* stream) {
  if (n < 0 || n_rows < 0 || n >= ((int64_t)1 << 31)) return (int)hipErrorInvalidValue;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) return gn_seg_offsets_i32(keys, 0, n_rows, seg_off, stream);
  if (!ws || ws_bytes < gn_csr_ws_bytes(n, n_rows)) return (int)hipErrorInvalidValue;
  const unsigned bits = key_bits(n_rows);
  int32_t* sorted = static_cast<int32_t*>(ws);
  char* temp = static_cast<char*>(ws) + align256((size_t)n * sizeof(int32_t));
  size_t temp_bytes = (size_t)ws_bytes - align256((size_t)n * sizeof(int32_t));
  // stable: equal keys keep their input order — the permutation of torch.argsort(keys, stable=True)
  const hipError_t e = use_onesweep(n) ? sort_pairs<cfg_onesweep>(temp, temp_bytes, keys, sorted, perm, n, bits, st)
                                       : sort_pairs<cfg_default>(temp, temp_bytes, keys, sorted, perm, n, bits, st);
  if (e != hipSuccess) return (int)e;
  return gn_seg_offsets_i32(sorted, n, n_rows, seg_off, stream);
}
