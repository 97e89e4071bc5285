/* gemnet_index.h — C ABI of the host-side graph/index builder (P12 of SURVEY.md §8a).
 *
 * Replaces DataContainer.__getitem__'s index construction
 * (/root/reference/gemnet/training/data_container.py:244-408: per-molecule dense distance matrix,
 * scipy CSR adjacency, get_triplets :410-425, get_quadruplets :427-489 and the numba helpers
 * repeat_blocks :520-546 / ragged_range :548-565) with a single-pass C++ builder.
 *
 * Integer semantics are those of the reference; the ORDER inside one reduce segment is canonical
 * (ascending expand edge) where the reference's depends on numpy's unstable argsort (:326,:371).
 * Distances are evaluated in the dtype of R exactly as numpy does for float32/float64 input
 * (sum of squares left to right, one rounding per operation, `<=` against the cutoff).
 *
 * Host only (no HIP): compiled with g++ into gemnet_pytorch_amd/csrc/libgemnet_index.so.
 * Thread-safe, no global state; the handle owns its arrays until gn_index_free().
 */
#ifndef GEMNET_INDEX_H
#define GEMNET_INDEX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gn_index gn_index;

/* R: (sum N, 3) row-major, float32 (r_is_f64 = 0) or float64 (r_is_f64 = 1); N: atoms per molecule (B,).
 * Returns NULL on allocation failure / invalid arguments. */
gn_index* gn_index_build(const void* R, int r_is_f64, const int32_t* N, int B, double cutoff,
                         double int_cutoff, int triplets_only);
void gn_index_free(gn_index* h);

/* Number of elements of the named array ("batch_seg", "id_undir", "id_swap", "id_c", "id_a",
 * "id3_expand_ba", "id3_reduce_ca", "Kidx3", "id4_int_b", "id4_int_a", "id4_reduce_ca",
 * "id4_expand_db", "id4_reduce_cab", "id4_expand_abd", "Kidx4", "id4_reduce_intm_ca",
 * "id4_expand_intm_db", "id4_reduce_intm_ab", "id4_expand_intm_ab"); -1 for an unknown key. */
int64_t gn_index_size(const gn_index* h, const char* key);
/* Copy the named array into out (int64, caller-owned, gn_index_size elements). 0 on success. */
int gn_index_copy(const gn_index* h, const char* key, int64_t* out);

/* The reference's numba helpers, exposed for their docstring known-answer tests. */
int64_t gn_repeat_blocks(const int64_t* sizes, const int64_t* repeats, int n, int64_t* out, int64_t cap);
int64_t gn_ragged_range(const int64_t* sizes, int n, int64_t* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
