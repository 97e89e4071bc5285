// Host-side graph/index builder (include/gemnet_index.h).  C++17, no dependencies.
//
// What the reference does with numpy + scipy.sparse + numba per batch
// (gemnet/training/data_container.py:244-408,410-489,520-565), as explicit loops over CSR
// neighbour lists.  Output order is canonical (see header).
#include "../../include/gemnet_index.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct gn_index {
  std::map<std::string, std::vector<int64_t>> arr;
};

namespace {

using vec = std::vector<int64_t>;

// ||R_i - R_j|| <= cutoff evaluated like np.linalg.norm(R[:,None]-R[None,:], axis=-1) <= cutoff in
// dtype T: d = R_i - R_j; s = (d0*d0 + d1*d1) + d2*d2, each op rounded to T; sqrt in T.
template <typename T>
inline bool within(const T* Ri, const T* Rj, T cutoff) {
  volatile T d0 = Ri[0] - Rj[0], d1 = Ri[1] - Rj[1], d2 = Ri[2] - Rj[2];
  volatile T q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
  volatile T s = q0 + q1;
  s = s + q2;
  volatile T d = std::sqrt((T)s);
  return d <= cutoff;
}

// row-major list of (t, s), t != s, within cutoff, per molecule with global atom offsets
template <typename T>
void pairs(const T* R, const int32_t* N, int B, double cutoff, vec& t_out, vec& s_out) {
  int64_t off = 0;
  const T c = (T)cutoff;
  for (int b = 0; b < B; ++b) {
    const int n = N[b];
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        if (i != j && within<T>(R + 3 * (off + i), R + 3 * (off + j), c)) {
          t_out.push_back(off + i);
          s_out.push_back(off + j);
        }
    off += n;
  }
}

void kidx(const vec& sorted_ids, vec& out) {
  out.resize(sorted_ids.size());
  int64_t k = 0;
  for (size_t i = 0; i < sorted_ids.size(); ++i) {
    k = (i > 0 && sorted_ids[i] == sorted_ids[i - 1]) ? k + 1 : 0;
    out[i] = k;
  }
}

template <typename T>
gn_index* build(const T* R, const int32_t* N, int B, double cutoff, double int_cutoff, int triplets_only) {
  auto* h = new gn_index();
  auto& A = h->arr;
  int64_t nAtoms = 0;
  vec& batch_seg = A["batch_seg"];
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < N[b]; ++i) batch_seg.push_back(b);
    nAtoms += N[b];
  }
  static const char* keysT[] = {"id_undir", "id_swap", "id_c", "id_a", "id3_expand_ba", "id3_reduce_ca", "Kidx3"};
  static const char* keysQ[] = {"id4_int_b", "id4_int_a", "id4_reduce_ca", "id4_expand_db", "id4_reduce_cab",
                                "id4_expand_abd", "Kidx4", "id4_reduce_intm_ca", "id4_expand_intm_db",
                                "id4_reduce_intm_ab", "id4_expand_intm_ab"};
  for (auto k : keysT) A[k];
  if (!triplets_only)
    for (auto k : keysQ) A[k];

  vec pt, ps;
  pairs<T>(R, N, B, cutoff, pt, ps);
  if (pt.empty()) return h;  // no-edge early return (data_container.py:282-285)

  // undirected edges once (t < s) in row-major order, then the reversed list (:289-293)
  vec& id_a = A["id_a"];
  vec& id_c = A["id_c"];
  for (size_t k = 0; k < pt.size(); ++k)
    if (pt[k] < ps[k]) { id_a.push_back(pt[k]); id_c.push_back(ps[k]); }
  const int64_t half = (int64_t)id_a.size();
  for (int64_t k = 0; k < half; ++k) { id_a.push_back(id_c[k]); id_c.push_back(id_a[k]); }
  const int64_t E = 2 * half;
  vec& id_undir = A["id_undir"];
  vec& id_swap = A["id_swap"];
  id_undir.resize(E);
  id_swap.resize(E);
  for (int64_t k = 0; k < half; ++k) {
    id_undir[k] = id_undir[k + half] = k;
    id_swap[k] = k + half;
    id_swap[k + half] = k;
  }

  // incoming edges of every atom, ordered by source atom (scipy canonical CSR column order)
  std::vector<int64_t> in_ptr(nAtoms + 1, 0);
  for (int64_t e = 0; e < E; ++e) in_ptr[id_a[e] + 1]++;
  for (int64_t a = 0; a < nAtoms; ++a) in_ptr[a + 1] += in_ptr[a];
  vec in_edge(E);
  {
    std::vector<int64_t> fill(in_ptr.begin(), in_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) in_edge[fill[id_a[e]]++] = e;
    for (int64_t a = 0; a < nAtoms; ++a)
      std::sort(in_edge.begin() + in_ptr[a], in_edge.begin() + in_ptr[a + 1],
                [&](int64_t x, int64_t y) { return id_c[x] < id_c[y]; });
  }

  // triplets: reduce edge r = (c->a); expand edges x = (b->a), b != c, ascending x
  {
    vec& red = A["id3_reduce_ca"];
    vec& exp = A["id3_expand_ba"];
    vec tmp;
    for (int64_t r = 0; r < E; ++r) {
      const int64_t a = id_a[r];
      tmp.assign(in_edge.begin() + in_ptr[a], in_edge.begin() + in_ptr[a + 1]);
      std::sort(tmp.begin(), tmp.end());
      for (int64_t x : tmp)
        if (id_c[x] != id_c[r]) { red.push_back(r); exp.push_back(x); }
    }
    kidx(red, A["Kidx3"]);
  }
  if (triplets_only) return h;

  // interaction edges (both directions, row-major (a, b)) and quadruplets (:427-489)
  vec& int_a = A["id4_int_a"];
  vec& int_b = A["id4_int_b"];
  pairs<T>(R, N, B, int_cutoff, int_a, int_b);
  vec& red_intm_ca = A["id4_reduce_intm_ca"];
  vec& exp_intm_db = A["id4_expand_intm_db"];
  vec& red_intm_ab = A["id4_reduce_intm_ab"];
  vec& exp_intm_ab = A["id4_expand_intm_ab"];
  const int64_t nInt = (int64_t)int_a.size();
  std::vector<int64_t> start_t(nInt + 1, 0), start_s(nInt + 1, 0);
  for (int64_t q = 0; q < nInt; ++q) {
    const int64_t a = int_a[q], b = int_b[q];
    for (int64_t k = in_ptr[a]; k < in_ptr[a + 1]; ++k) { red_intm_ca.push_back(in_edge[k]); red_intm_ab.push_back(q); }
    for (int64_t k = in_ptr[b]; k < in_ptr[b + 1]; ++k) { exp_intm_db.push_back(in_edge[k]); exp_intm_ab.push_back(q); }
    start_t[q + 1] = (int64_t)red_intm_ca.size();
    start_s[q + 1] = (int64_t)exp_intm_db.size();
  }
  struct Quad { int64_t rca, xdb, rcab, xabd; };
  std::vector<Quad> quads;
  for (int64_t q = 0; q < nInt; ++q)
    for (int64_t j = start_s[q]; j < start_s[q + 1]; ++j)      // in-edges of b (outer, :451-455)
      for (int64_t i = start_t[q]; i < start_t[q + 1]; ++i) {  // in-edges of a (inner, repeat_blocks :446)
        const int64_t rca = red_intm_ca[i], xdb = exp_intm_db[j];
        const int64_t c = id_c[rca], a = id_a[rca], b = id_a[xdb], d = id_c[xdb];
        if (c != b && a != d && c != d) quads.push_back({rca, xdb, i, j});
      }
  std::sort(quads.begin(), quads.end(), [](const Quad& x, const Quad& y) {
    return x.rca != y.rca ? x.rca < y.rca : x.xdb < y.xdb;
  });
  vec& q_rca = A["id4_reduce_ca"];
  vec& q_xdb = A["id4_expand_db"];
  vec& q_rcab = A["id4_reduce_cab"];
  vec& q_xabd = A["id4_expand_abd"];
  q_rca.reserve(quads.size()); q_xdb.reserve(quads.size());
  q_rcab.reserve(quads.size()); q_xabd.reserve(quads.size());
  for (const Quad& q : quads) {
    q_rca.push_back(q.rca); q_xdb.push_back(q.xdb); q_rcab.push_back(q.rcab); q_xabd.push_back(q.xabd);
  }
  kidx(q_rca, A["Kidx4"]);
  return h;
}

}  // namespace

extern "C" gn_index* gn_index_build(const void* R, int r_is_f64, const int32_t* N, int B, double cutoff,
                                    double int_cutoff, int triplets_only) {
  if (B < 0 || (B > 0 && (!R || !N))) return nullptr;
  try {
    if (r_is_f64) return build<double>(static_cast<const double*>(R), N, B, cutoff, int_cutoff, triplets_only);
    return build<float>(static_cast<const float*>(R), N, B, cutoff, int_cutoff, triplets_only);
  } catch (...) {
    return nullptr;
  }
}

extern "C" void gn_index_free(gn_index* h) { delete h; }

extern "C" int64_t gn_index_size(const gn_index* h, const char* key) {
  if (!h) return -1;
  auto it = h->arr.find(key);
  return it == h->arr.end() ? -1 : (int64_t)it->second.size();
}

extern "C" int gn_index_copy(const gn_index* h, const char* key, int64_t* out) {
  if (!h) return 1;
  auto it = h->arr.find(key);
  if (it == h->arr.end()) return 2;
  if (!it->second.empty()) std::memcpy(out, it->second.data(), it->second.size() * sizeof(int64_t));
  return 0;
}

extern "C" int64_t gn_repeat_blocks(const int64_t* sizes, const int64_t* repeats, int n, int64_t* out, int64_t cap) {
  int64_t start = 0, oi = 0;
  for (int i = 0; i < n; ++i) {
    for (int64_t r = 0; r < repeats[i]; ++r)
      for (int64_t k = 0; k < sizes[i]; ++k) {
        if (oi < cap) out[oi] = start + k;
        ++oi;
      }
    start += sizes[i];
  }
  return oi;
}

extern "C" int64_t gn_ragged_range(const int64_t* sizes, int n, int64_t* out, int64_t cap) {
  int64_t oi = 0;
  for (int i = 0; i < n; ++i)
    for (int64_t k = 0; k < sizes[i]; ++k) {
      if (oi < cap) out[oi] = k;
      ++oi;
    }
  return oi;
}
