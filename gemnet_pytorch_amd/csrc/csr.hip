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

}  // namespace

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
