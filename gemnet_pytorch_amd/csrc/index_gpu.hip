// Device-resident index construction: what DataContainer.__getitem__ does on the host with
// numpy + scipy.sparse + numba per batch (gemnet/training/data_container.py:244-408 edges / id_swap / id_undir,
// :410-425 triplets, :427-489 quadruplets, :520-565 repeat_blocks / ragged_range), rebuilt every MD step in
// ase_calculator.py:155-158.  Same output as the host builder csrc/index_build.cpp (canonical order: triplets
// and quadruplets sorted by (reduce edge, expand edge); the reference's own order inside a reduce segment
// depends on numpy's unstable argsort), bit-exact integers.
//
// Integer / byte work, bound by HBM writes of the quadruplet arrays (5 x int32 x Q).  Molecules are dense
// (the reference builds the full n x n distance matrix per molecule, :255-258), so adjacency lives in per-
// molecule n x n byte / int32 matrices; all ragged outputs are sized by a count pass + exclusive scan and
// written by ballot-compaction (one wave per reduce edge walks its candidates in canonical order), so the
// order is deterministic and no atomics are used.
//
// Distances are evaluated exactly like np.linalg.norm(R[:,None]-R[None,:], axis=-1) <= cutoff in R's dtype:
// d = R_i - R_j, s = (d0*d0 + d1*d1) + d2*d2 with every operation rounded (no FMA contraction), correctly
// rounded sqrt, comparison against the cutoff cast to that dtype.
#include "common.h"

namespace {

template <typename T> struct RN;
template <> struct RN<float> {
  static __device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
  static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
  static __device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }
  static __device__ __forceinline__ float sqrt(float a) { return __fsqrt_rn(a); }
};
template <> struct RN<double> {
  static __device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
  static __device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }
  static __device__ __forceinline__ double sub(double a, double b) { return __dsub_rn(a, b); }
  static __device__ __forceinline__ double sqrt(double a) { return __dsqrt_rn(a); }
};

template <typename T>
__device__ __forceinline__ T dist(const T* __restrict__ R, int i, int j) {
  const T d0 = RN<T>::sub(R[3 * i], R[3 * j]);
  const T d1 = RN<T>::sub(R[3 * i + 1], R[3 * j + 1]);
  const T d2 = RN<T>::sub(R[3 * i + 2], R[3 * j + 2]);
  const T s = RN<T>::add(RN<T>::add(RN<T>::mul(d0, d0), RN<T>::mul(d1, d1)), RN<T>::mul(d2, d2));
  return RN<T>::sqrt(s);
}

// adj / iadj: per molecule m an n x n byte matrix at sq_off[m]; grid (ceil(nmax^2 / 256), B)
template <typename T>
__global__ void idx_adj_kernel(const T* __restrict__ R, const int32_t* __restrict__ mol_off,
                               const int32_t* __restrict__ sq_off, T cutoff, T int_cutoff, int quad,
                               uint8_t* __restrict__ adj, uint8_t* __restrict__ iadj) {
  const int m = blockIdx.y;
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n * n) return;
  const int x = p / n, y = p - x * n;
  uint8_t w = 0, wi = 0;
  if (x != y) {
    const T d = dist<T>(R, a0 + x, a0 + y);
    w = d <= cutoff;
    wi = d <= int_cutoff;
  }
  adj[sq_off[m] + p] = w;
  if (quad) iadj[sq_off[m] + p] = wi;
}

// per atom: degree, number of upper neighbours, interaction degree, molecule id
__global__ void idx_deg_kernel(const int32_t* __restrict__ mol_off, const int32_t* __restrict__ sq_off,
                               const int32_t* __restrict__ atom_mol, int A, int quad,
                               const uint8_t* __restrict__ adj, const uint8_t* __restrict__ iadj,
                               int32_t* __restrict__ deg, int32_t* __restrict__ up, int32_t* __restrict__ ideg) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A) return;
  const int m = atom_mol[g];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, x = g - a0;
  const uint8_t* __restrict__ row = adj + sq_off[m] + (size_t)x * n;
  int d = 0, u = 0, di = 0;
  for (int y = 0; y < n; ++y) {
    d += row[y];
    u += (y > x) & row[y];
  }
  if (quad) {
    const uint8_t* __restrict__ irow = iadj + sq_off[m] + (size_t)x * n;
    for (int y = 0; y < n; ++y) di += irow[y];
  }
  deg[g] = d;
  up[g] = u;
  if (quad) ideg[g] = di;
}

// single-block exclusive scan of n int32 (n up to a few million: n/1024 sequential chunks); out[n] = total
__global__ __launch_bounds__(1024) void idx_scan_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                        int64_t n, int* __restrict__ overflow) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int64_t v = i < n ? (int64_t)in[i] : 0;
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
    const int64_t excl = carry + woff + s - v;
    if (i < n) {
      if (excl > 0x7fffffffLL) *overflow = 1;
      out[i] = (int32_t)excl;
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + s;
    __syncthreads();
  }
  if (tid == 0) {
    if (carry_s > 0x7fffffffLL) *overflow = 1;
    out[n] = (int32_t)carry_s;
  }
}

// edges: pair (t < s) -> e = off_half[t] + k and e + H; M[x][y] = id of the edge with target x, source y
__global__ void idx_edges_kernel(const int32_t* __restrict__ mol_off, const int32_t* __restrict__ sq_off,
                                 const int32_t* __restrict__ atom_mol, int A, int quad,
                                 const uint8_t* __restrict__ adj, const uint8_t* __restrict__ iadj,
                                 const int32_t* __restrict__ off_half, const int32_t* __restrict__ off_int,
                                 int32_t* __restrict__ id_a, int32_t* __restrict__ id_c,
                                 int32_t* __restrict__ id_undir, int32_t* __restrict__ id_swap,
                                 int32_t* __restrict__ int_a, int32_t* __restrict__ int_b,
                                 int32_t* __restrict__ Mx, int32_t* __restrict__ MI, int e_cap, int eint_cap) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A) return;
  const int m = atom_mol[g];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, x = g - a0;
  const int H = off_half[A];
  if (2 * (int64_t)H > e_cap) return;       // capacity forms (gn_index_gpu_padded_*): nothing is written, the step is flagged
  if (quad && off_int[A] > eint_cap) return;
  const size_t base = (size_t)sq_off[m];
  const uint8_t* __restrict__ row = adj + base + (size_t)x * n;
  int e = off_half[g];
  for (int y = x + 1; y < n; ++y) {
    if (row[y]) {
      id_a[e] = g; id_c[e] = a0 + y;
      id_a[e + H] = a0 + y; id_c[e + H] = g;
      id_undir[e] = e; id_undir[e + H] = e;
      id_swap[e] = e + H; id_swap[e + H] = e;
      Mx[base + (size_t)x * n + y] = e;
      Mx[base + (size_t)y * n + x] = e + H;
      ++e;
    }
  }
  if (quad) {
    const uint8_t* __restrict__ irow = iadj + base + (size_t)x * n;
    int i = off_int[g];
    for (int y = 0; y < n; ++y) {
      if (irow[y]) {
        int_a[i] = g; int_b[i] = a0 + y;
        MI[base + (size_t)x * n + y] = i;
        ++i;
      }
    }
  }
}

// incoming edges of every atom ordered by source atom (scipy canonical CSR column order) + position of each edge
__global__ void idx_in_kernel(const int32_t* __restrict__ mol_off, const int32_t* __restrict__ sq_off,
                              const int32_t* __restrict__ atom_mol, int A, const uint8_t* __restrict__ adj,
                              const int32_t* __restrict__ Mx, const int32_t* __restrict__ in_ptr,
                              int32_t* __restrict__ in_edge, int32_t* __restrict__ pos_in,
                              const int32_t* __restrict__ half_total, int e_cap) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A) return;
  if (half_total && 2 * (int64_t)*half_total > e_cap) return;
  const int m = atom_mol[g];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, x = g - a0;
  const size_t base = (size_t)sq_off[m] + (size_t)x * n;
  int k = 0;
  const int p0 = in_ptr[g];
  for (int y = 0; y < n; ++y) {
    if (adj[base + y]) {
      const int e = Mx[base + y];
      in_edge[p0 + k] = e;
      pos_in[e] = k;
      ++k;
    }
  }
}

__global__ void idx_cnt3_kernel(const int32_t* __restrict__ id_a, const int32_t* __restrict__ deg, int E,
                                int32_t* __restrict__ cnt3) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < E) cnt3[r] = deg[id_a[r]] - 1;
}

// capacity form: the number of edges lives on the device (2 * *half_total); rows behind it count zero triplets
__global__ void idx_cnt3_cap_kernel(const int32_t* __restrict__ id_a, const int32_t* __restrict__ deg,
                                    const int32_t* __restrict__ half_total, int n, int e_cap, int32_t* __restrict__ cnt3) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t E = 2 * (int64_t)*half_total;
  cnt3[r] = (E <= e_cap && r < E) ? deg[id_a[r]] - 1 : 0;
}

// one wave per reduce edge r = (c -> a): expand edges x = (b -> a), b != c, ascending edge id
// = sources b > a ascending (first-half ids), then sources b < a ascending (second-half ids)
__global__ __launch_bounds__(256) void idx_trip_kernel(const int32_t* __restrict__ mol_off,
                                                       const int32_t* __restrict__ sq_off,
                                                       const int32_t* __restrict__ atom_mol,
                                                       const int32_t* __restrict__ id_a, const int32_t* __restrict__ id_c,
                                                       int E, const uint8_t* __restrict__ adj,
                                                       const int32_t* __restrict__ Mx, const int32_t* __restrict__ off3,
                                                       int32_t* __restrict__ red, int32_t* __restrict__ exp,
                                                       int32_t* __restrict__ kidx,
                                                       const int32_t* __restrict__ half_total, int e_cap, int t_cap,
                                                       const int32_t* __restrict__ skip) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (skip && skip[3]) return;       // (the decide kernel of gn_index_gpu_padded_q found that this batch does not fit)
  if (half_total) {      // capacity form: E on the device, E = the grid's upper bound; nothing is written past the capacities
    const int64_t Ed = 2 * (int64_t)*half_total;
    if (Ed > e_cap || r >= Ed || off3[E] > t_cap) return;
  } else if (r >= E) {
    return;
  }
  const int ga = id_a[r], gc = id_c[r];
  const int m = atom_mol[ga];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, a = ga - a0, c = gc - a0;
  const size_t base = (size_t)sq_off[m] + (size_t)a * n;
  int o = off3[r];
  const int o0 = o;
  // candidate index j in [0, n-1): b = a+1+j for j < n-1-a, else b = j - (n-1-a)
  const int nup = n - 1 - a;
  for (int j0 = 0; j0 < n - 1; j0 += 64) {
    const int j = j0 + lane;
    bool ok = false;
    int b = 0;
    if (j < n - 1) {
      b = j < nup ? a + 1 + j : j - nup;
      ok = adj[base + b] && b != c;
    }
    const uint64_t mask = __ballot(ok);
    if (ok) {
      const int w = o + __popcll(mask & ((1ull << lane) - 1ull));
      red[w] = r;
      exp[w] = Mx[base + b];
      if (kidx) kidx[w] = w - o0;
    }
    o += __popcll(mask);
  }
}

// Commit + padding of the capacity form.  The build above wrote E edges / T triplets (counts on the device) into STAGING
// arrays; this kernel copies them into the arrays the captured model reads and fills the rows behind them with the dummy
// molecule's pad rows (padded.py: _pad_edges / _pad_triplets, triplets-only layout: groups of 3 dummy atoms behind atom
// a_cap, pad edges in quads b->a, a->b, c->a, a->c).  A batch that does not fit (or whose padding breaks the rules of
// padded.py) leaves the arrays of the PREVIOUS step in place — always valid indices — and reports through state[]:
//   state[0] |= err (sticky), state[1] = E, state[2] = T, state[3] = err of this step
//   err: 1 E > e_cap, 2 T > t_cap, 4 pad triplets without a complete quad of pad edges, 8 dummy in-degree above deg_bound
struct PadT {
  const int32_t *s_c, *s_a, *s_swap, *s_undir, *s_red, *s_exp;
  int32_t *id_c, *id_a, *id_swap, *id_undir, *red, *exp;
};

__global__ void idx_commit_pad_t_kernel(PadT p, const int32_t* __restrict__ half_total, const int32_t* __restrict__ t_total,
                                        int e_cap, int t_cap, int a_cap, int G, int deg_bound, int32_t* __restrict__ state) {
  const int64_t E = 2 * (int64_t)*half_total, T = E <= e_cap ? (int64_t)*t_total : 0;
  int err = 0;
  if (E > e_cap) err |= 1;
  if (T > t_cap) err |= 2;
  const int64_t ep = e_cap - E, tp = t_cap - T;
  if (!err) {
    if ((tp > 0 && ep < 4) || (tp & 1)) err |= 4;
    if ((ep / 2 + G - 1) / G > (deg_bound > 2 ? deg_bound : 2)) err |= 8;
  }
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  if (tid == 0) {
    state[0] |= err;
    state[1] = (int32_t)(E > 0x7fffffff ? 0x7fffffff : E);
    state[2] = (int32_t)T;
    state[3] = err;
  }
  if (err) return;
  for (int64_t i = tid; i < e_cap; i += nth) {
    if (i < E) {
      p.id_c[i] = p.s_c[i]; p.id_a[i] = p.s_a[i]; p.id_swap[i] = p.s_swap[i]; p.id_undir[i] = p.s_undir[i];
    } else {
      const int64_t k = i - E, pair = k >> 1;
      const int rev = (int)(k & 1), typ = (int)(pair & 1), grp = (int)((pair >> 1) % G);
      const int a = a_cap + 3 * grp, other = a + 1 + typ;
      p.id_c[i] = rev ? a : other;
      p.id_a[i] = rev ? other : a;
      p.id_swap[i] = (int32_t)(E + (k ^ 1));
      p.id_undir[i] = (int32_t)(E / 2 + pair);
    }
  }
  const int64_t n_fwd = 2 * (ep / 4), den = tp > 1 ? tp : 1;
  for (int64_t i = tid; i < t_cap; i += nth) {
    if (i < T) {
      p.red[i] = p.s_red[i]; p.exp[i] = p.s_exp[i];
    } else {
      const int64_t f = ((i - T) * n_fwd) / den;
      p.red[i] = (int32_t)(E + 2 * f);
      p.exp[i] = (int32_t)(E + 2 * (f ^ 1));
    }
  }
}

// ---- capacity form for quadruplet models (gn_index_gpu_padded_q) ------------------------------------------------------------
// All five counts are known on the device before any of the large arrays is written: `idx_decide_q_kernel` (one thread)
// compares them with the capacities and the rules of the padding scheme (padded.py: _fill / _check_quad_padding /
// pad_in_degree, quadruplet layout) and writes the verdict into state[3]; the triplet / intermediate-triplet / quadruplet
// writers then go STRAIGHT into the arrays the model reads (they return at once when the verdict is bad, so those arrays
// keep the previous step's valid contents), and the commit kernel copies the two small staged families and writes every
// family's pad rows.   state (int32[8]): [0] |= err (sticky)  [1] E  [2] T  [3] err of this call  [4] Eint  [5] I  [6] Q
//   err bits: 1 E, 2 T, 16 Eint, 32 I, 64 Q over capacity; 4 pad rows without the pad rows they refer to; 8 dummy in-degree;
//   128 the two intermediate-triplet lists differ in length (cannot happen for symmetric interaction lists)
struct QCounts { const int32_t *half_total, *t_total, *eint_total, *ica_total, *idb_total, *q_total; };
struct QCaps { int e, t, eint, i, q, a, G, deg_bound; };

__device__ __forceinline__ int q_decide(const QCounts& c, const QCaps& k, int64_t* n) {
  const int64_t E = 2 * (int64_t)*c.half_total, T = *c.t_total, Ei = *c.eint_total, Ica = *c.ica_total, Idb = *c.idb_total,
                Q = *c.q_total;
  n[0] = E; n[1] = T; n[2] = Ei; n[3] = Ica; n[4] = Q;
  int err = 0;
  if (E > k.e) err |= 1;
  if (Ei > k.eint) err |= 16;
  if (err) return err;                 // (the counts below were taken from rows that were not written)
  if (T > k.t) err |= 2;
  if (Ica > k.i) err |= 32;
  if (Q > k.q) err |= 64;
  if (Ica != Idb) err |= 128;
  if (err) return err;
  const int64_t ep = k.e - E, tp = k.t - T, eintp = k.eint - Ei, ip = k.i - Ica, qp = k.q - Q;
  if ((tp & 1) || ((tp || ip || qp) && ep < 6) || (ip && eintp < 1) || (qp && ip < 1)) err |= 4;
  const int64_t units = (ep + 5) / 6;
  if (2 * ((units + k.G - 1) / k.G) > (k.deg_bound > 2 ? k.deg_bound : 2)) err |= 8;
  return err;
}

__global__ void idx_decide_q_kernel(QCounts c, QCaps k, int32_t* __restrict__ state) {
  if (blockIdx.x || threadIdx.x) return;
  int64_t n[5];
  const int err = q_decide(c, k, n);
  auto clip = [](int64_t v) { return (int32_t)(v > 0x7fffffff ? 0x7fffffff : v); };
  state[0] |= err;
  state[1] = clip(n[0]); state[2] = clip(n[1]); state[3] = err; state[4] = clip(n[2]); state[5] = clip(n[3]); state[6] = clip(n[4]);
}

struct PadQ {
  const int32_t *s_c, *s_a, *s_swap, *s_undir, *s_int_a, *s_int_b;
  int32_t *id_c, *id_a, *id_swap, *id_undir, *red3, *exp3, *int_a, *int_b, *intm_ca, *intm_db, *intm_red_ab, *intm_exp_ab,
      *q_red_ca, *q_exp_db, *q_red_cab, *q_exp_abd;
};

__global__ void idx_commit_pad_q_kernel(PadQ p, QCaps k, const int32_t* __restrict__ state) {
  if (state[3]) return;
  const int64_t E = state[1], T = state[2], Ei = state[4], I = state[5], Q = state[6];
  const int64_t ep = k.e - E, tp = k.t - T, eintp = k.eint - Ei, ip = k.i - I, qp = k.q - Q;
  const int G = k.G;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  // edges: copy + pad (units of six: b->a, a->b, c->a, a->c, d->b, b->d over groups of four dummy atoms)
  for (int64_t i = tid; i < k.e; i += nth) {
    if (i < E) {
      p.id_c[i] = p.s_c[i]; p.id_a[i] = p.s_a[i]; p.id_swap[i] = p.s_swap[i]; p.id_undir[i] = p.s_undir[i];
    } else {
      const int64_t kk = i - E, pair = kk >> 1;
      const int rev = (int)(kk & 1), typ = (int)(pair % 3), grp = (int)((pair / 3) % G);
      const int a = k.a + 4 * grp;
      const int src = typ == 2 ? a + 3 : a + 1 + typ, dst = typ == 2 ? a + 1 : a;
      p.id_c[i] = rev ? dst : src;
      p.id_a[i] = rev ? src : dst;
      p.id_swap[i] = (int32_t)(E + (kk ^ 1));
      p.id_undir[i] = (int32_t)(E / 2 + pair);
    }
  }
  // pad triplets: the forward edges b->a, c->a (offsets 0 and 2) of the complete units
  {
    const int64_t nf = 2 * (ep / 6), den = tp > 1 ? tp : 1;
    for (int64_t j = tid; j < tp; j += nth) {
      const int64_t sidx = (j * nf) / den, u = sidx >> 1, w = sidx & 1;
      p.red3[T + j] = (int32_t)(E + 6 * u + 2 * w);
      p.exp3[T + j] = (int32_t)(E + 6 * u + 2 * (1 - w));
    }
  }
  // interaction edges: copy + pad (pairs b->a, a->b cycling over the groups)
  for (int64_t i = tid; i < k.eint; i += nth) {
    if (i < Ei) {
      p.int_a[i] = p.s_int_a[i]; p.int_b[i] = p.s_int_b[i];
    } else {
      const int64_t kk = i - Ei, pr = kk >> 1;
      const int rev = (int)(kk & 1), a = k.a + 4 * (int)(pr % G);
      p.int_b[i] = rev ? a : a + 1;
      p.int_a[i] = rev ? a + 1 : a;
    }
  }
  // pad intermediate triplets / quadruplets
  const int64_t n_ab = (eintp + 1) / 2, n_units = ep / 6 > 1 ? ep / 6 : 1, iden = ip > 1 ? ip : 1;
  auto intm_unit = [&](int64_t i, int64_t& pp) {     // unit of pad intermediate triplet i, and its interaction-edge pair
    pp = (i * n_ab) / iden;
    const int64_t g = pp % G;
    return g < n_units ? g : pp % n_units;
  };
  for (int64_t i = tid; i < ip; i += nth) {
    int64_t pp;
    const int64_t u = intm_unit(i, pp);
    p.intm_ca[I + i] = (int32_t)(E + 6 * u + 2);
    p.intm_db[I + i] = (int32_t)(E + 6 * u + 4);
    p.intm_red_ab[I + i] = (int32_t)(Ei + 2 * pp);
    p.intm_exp_ab[I + i] = (int32_t)(Ei + 2 * pp);
  }
  {
    const int64_t qden = qp > 1 ? qp : 1, imod = ip > 1 ? ip : 1;
    for (int64_t q = tid; q < qp; q += nth) {
      const int64_t u = (q * n_units) / qden, im = q % imod;
      int64_t pp;
      const int64_t ui = intm_unit(im, pp);
      p.q_red_ca[Q + q] = (int32_t)(E + 6 * u + 2);
      p.q_exp_abd[Q + q] = (int32_t)(I + im);
      p.q_red_cab[Q + q] = (int32_t)(I + im);
      p.q_exp_db[Q + q] = (int32_t)(E + 6 * ui + 4);
    }
  }
}

// x[i] = NaN for all i when state[3] (the error of this step's index build) is set: a step whose batch did not fit must not
// hand out the numbers computed from the previous step's arrays
__global__ void idx_poison_kernel(float* __restrict__ x, int64_t n, const int32_t* __restrict__ state) {
  if (state[3] == 0) return;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = __builtin_nanf("");
}

__global__ void idx_cnt_intm_kernel(const int32_t* __restrict__ int_a, const int32_t* __restrict__ int_b,
                                    const int32_t* __restrict__ deg, int Eint, int32_t* __restrict__ cnt_ca,
                                    int32_t* __restrict__ cnt_db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Eint) {
    cnt_ca[i] = deg[int_a[i]];
    cnt_db[i] = deg[int_b[i]];
  }
}

// capacity form: the number of interaction edges lives on the device; rows behind it count zero intermediate triplets
__global__ void idx_cnt_intm_cap_kernel(const int32_t* __restrict__ int_a, const int32_t* __restrict__ int_b,
                                        const int32_t* __restrict__ deg, const int32_t* __restrict__ eint_total, int n,
                                        int eint_cap, const int32_t* __restrict__ half_total, int e_cap,
                                        int32_t* __restrict__ cnt_ca, int32_t* __restrict__ cnt_db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int Ei = *eint_total;
  const bool ok = Ei <= eint_cap && 2 * (int64_t)*half_total <= e_cap && i < Ei;   // (the edge kernel wrote its staging arrays)
  cnt_ca[i] = ok ? deg[int_a[i]] : 0;
  cnt_db[i] = ok ? deg[int_b[i]] : 0;
}

__global__ void idx_intm_kernel(const int32_t* __restrict__ int_a, const int32_t* __restrict__ int_b, int Eint,
                                const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_edge,
                                const int32_t* __restrict__ off_ca, const int32_t* __restrict__ off_db,
                                int32_t* __restrict__ red_intm_ca, int32_t* __restrict__ red_intm_ab,
                                int32_t* __restrict__ exp_intm_db, int32_t* __restrict__ exp_intm_ab,
                                const int32_t* __restrict__ eint_total, const int32_t* __restrict__ skip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (skip && skip[3]) return;
  if (eint_total) Eint = *eint_total;
  if (i >= Eint) return;
  const int a = int_a[i], b = int_b[i];
  int p = in_ptr[a], q = off_ca[i];
  for (int k = in_ptr[a + 1] - p; k > 0; --k, ++p, ++q) { red_intm_ca[q] = in_edge[p]; red_intm_ab[q] = i; }
  p = in_ptr[b]; q = off_db[i];
  for (int k = in_ptr[b + 1] - p; k > 0; --k, ++p, ++q) { exp_intm_db[q] = in_edge[p]; exp_intm_ab[q] = i; }
}

// quadruplets of reduce edge r = (c -> a): all (b, d) with b in intN(a), d in N(b), c != b, a != d, c != d
__global__ void idx_cnt4_kernel(const int32_t* __restrict__ mol_off, const int32_t* __restrict__ sq_off,
                                const int32_t* __restrict__ atom_mol, const int32_t* __restrict__ id_a,
                                const int32_t* __restrict__ id_c, int E, const uint8_t* __restrict__ adj,
                                const uint8_t* __restrict__ iadj, const int32_t* __restrict__ deg,
                                int32_t* __restrict__ cnt4, const int32_t* __restrict__ half_total, int e_cap) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= E) return;
  if (half_total) {       // capacity form: E = rows to write, the edge count on the device; rows behind it count zero
    const int64_t Ed = 2 * (int64_t)*half_total;
    if (Ed > e_cap || r >= Ed) { cnt4[r] = 0; return; }
  }
  const int ga = id_a[r], gc = id_c[r];
  const int m = atom_mol[ga];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, a = ga - a0, c = gc - a0;
  const size_t base = (size_t)sq_off[m];
  int cnt = 0;
  for (int b = 0; b < n; ++b)
    if (iadj[base + (size_t)a * n + b] && b != c)
      cnt += deg[a0 + b] - adj[base + (size_t)b * n + a] - adj[base + (size_t)b * n + c];
  cnt4[r] = cnt;
}

// one wave per reduce edge; expand edges (d -> b) in ascending edge id: first-half ids (b < d) are ordered by
// (b, d), second-half ids (b > d) by (d, b)
__global__ __launch_bounds__(256) void idx_quad_kernel(
    const int32_t* __restrict__ mol_off, const int32_t* __restrict__ sq_off, const int32_t* __restrict__ atom_mol,
    const int32_t* __restrict__ id_a, const int32_t* __restrict__ id_c, int E, const uint8_t* __restrict__ adj,
    const uint8_t* __restrict__ iadj, const int32_t* __restrict__ Mx, const int32_t* __restrict__ MI,
    const int32_t* __restrict__ pos_in, const int32_t* __restrict__ off_ca, const int32_t* __restrict__ off_db,
    const int32_t* __restrict__ off4, int32_t* __restrict__ red_ca, int32_t* __restrict__ exp_db,
    int32_t* __restrict__ red_cab, int32_t* __restrict__ exp_abd, int32_t* __restrict__ kidx,
    const int32_t* __restrict__ half_total, const int32_t* __restrict__ skip) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (skip && skip[3]) return;
  if (r >= E || (half_total && r >= 2 * (int64_t)*half_total)) return;
  const int ga = id_a[r], gc = id_c[r];
  const int m = atom_mol[ga];
  const int a0 = mol_off[m], n = mol_off[m + 1] - a0, a = ga - a0, c = gc - a0;
  const size_t base = (size_t)sq_off[m];
  const uint8_t* __restrict__ irow = iadj + base + (size_t)a * n;
  const int32_t* __restrict__ mirow = MI + base + (size_t)a * n;
  const int pr = pos_in[r];
  int o = off4[r];
  const int o0 = o;
  const int nn = n * n;
  for (int phase = 0; phase < 2; ++phase) {
    for (int j0 = 0; j0 < nn; j0 += 64) {
      const int j = j0 + lane;
      bool ok = false;
      int b = 0, d = 0;
      if (j < nn) {
        const int hi = j / n, lo = j - hi * n;      // phase 0: (b, d) = (hi, lo), b < d; phase 1: (d, b) = (hi, lo), b > d
        b = phase ? lo : hi;
        d = phase ? hi : lo;
        ok = hi < lo && irow[b] && b != c && d != a && d != c && adj[base + (size_t)b * n + d];
      }
      const uint64_t mask = __ballot(ok);
      if (ok) {
        const int w = o + __popcll(mask & ((1ull << lane) - 1ull));
        const int x = Mx[base + (size_t)b * n + d];   // edge with target b, source d
        const int i = mirow[b];                        // interaction edge (a, b)
        red_ca[w] = r;
        exp_db[w] = x;
        red_cab[w] = off_ca[i] + pr;
        exp_abd[w] = off_db[i] + pos_in[x];
        if (kidx) kidx[w] = w - o0;
      }
      o += __popcll(mask);
    }
  }
}

__global__ void idx_atom_mol_kernel(const int32_t* __restrict__ mol_off, int B, int32_t* __restrict__ atom_mol,
                                    int32_t* __restrict__ batch_seg) {
  const int m = blockIdx.x;
  for (int g = mol_off[m] + threadIdx.x; g < mol_off[m + 1]; g += blockDim.x) {
    atom_mol[g] = m;
    if (batch_seg) batch_seg[g] = m;
  }
}

inline size_t al(size_t x) { return (x + 63) & ~(size_t)63; }

}  // namespace

// workspace layout (bytes, 64-B aligned sections); see gn_index_gpu_ws_bytes
struct idx_ws {
  uint8_t *adj, *iadj;
  int32_t *Mx, *MI, *atom_mol, *deg, *up, *ideg, *off_half, *in_ptr, *off_int, *in_edge, *pos_in, *cnt, *off3, *off4,
      *cnt2, *off_ca, *off_db;
  int* overflow;
};

static size_t idx_layout(char* base, int A, int64_t sq, int64_t Emax, int64_t Eintmax, int quad, idx_ws* w) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += al(bytes); return p; };
  w->adj = (uint8_t*)take(sq);
  w->iadj = (uint8_t*)take(quad ? sq : 0);
  w->Mx = (int32_t*)take(4 * sq);
  w->MI = (int32_t*)take(quad ? 4 * sq : 0);
  w->atom_mol = (int32_t*)take(4 * (size_t)A);
  w->deg = (int32_t*)take(4 * (size_t)A);
  w->up = (int32_t*)take(4 * (size_t)A);
  w->ideg = (int32_t*)take(4 * (size_t)A);
  w->off_half = (int32_t*)take(4 * ((size_t)A + 1));
  w->in_ptr = (int32_t*)take(4 * ((size_t)A + 1));
  w->off_int = (int32_t*)take(4 * ((size_t)A + 1));
  w->in_edge = (int32_t*)take(4 * (size_t)Emax);
  w->pos_in = (int32_t*)take(4 * (size_t)Emax);
  w->cnt = (int32_t*)take(4 * (size_t)(Emax > Eintmax ? Emax : Eintmax));
  w->off3 = (int32_t*)take(4 * ((size_t)Emax + 1));
  w->off4 = (int32_t*)take(4 * ((size_t)Emax + 1));
  w->cnt2 = (int32_t*)take(4 * (size_t)Eintmax);
  w->off_ca = (int32_t*)take(4 * ((size_t)Eintmax + 1));
  w->off_db = (int32_t*)take(4 * ((size_t)Eintmax + 1));
  w->overflow = (int*)take(64);
  return o;
}

// Upper bounds used for the workspace: E <= sum n(n-1) = sq - A, Eint likewise.
extern "C" int64_t gn_index_gpu_ws_bytes(int A, int64_t sum_n2, int triplets_only) {
  idx_ws w;
  const int64_t emax = sum_n2 - A > 0 ? sum_n2 - A : 0;
  return (int64_t)idx_layout(nullptr, A, sum_n2, emax, triplets_only ? 0 : emax, !triplets_only, &w);
}

// Stage 1: adjacency, edges, incoming lists, counts.  sizes (host, 6 x int64) <- E, T, Eint, Ica, Idb, Q
// (synchronises the stream once to read them back).  Edge-level outputs are written here (caller sizes them with
// the upper bound E <= sum_n2 - A, or calls with NULL outputs first and again with buffers: idempotent).
extern "C" int gn_index_gpu_stage1(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B,
                                   int A, int nmax, int64_t sum_n2, double cutoff, double int_cutoff, int triplets_only,
                                   void* ws, int32_t* batch_seg, int32_t* id_a, int32_t* id_c, int32_t* id_undir,
                                   int32_t* id_swap, int32_t* int_a, int32_t* int_b, int64_t* sizes, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int quad = !triplets_only;
  for (int i = 0; i < 6; ++i) sizes[i] = 0;
  if (A <= 0 || B <= 0) return 0;
  if (sum_n2 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  idx_ws w;
  const int64_t emax = sum_n2 - A;
  idx_layout((char*)ws, A, sum_n2, emax, quad ? emax : 0, quad, &w);
  hipError_t e = hipMemsetAsync(w.overflow, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(idx_atom_mol_kernel, dim3(B), dim3(64), 0, st, mol_off, B, w.atom_mol, batch_seg);
  dim3 gadj(gn_cdiv((int64_t)nmax * nmax, 256), B);
  if (r_is_f64)
    hipLaunchKernelGGL((idx_adj_kernel<double>), gadj, dim3(256), 0, st, (const double*)R, mol_off, sq_off, cutoff,
                       int_cutoff, quad, w.adj, w.iadj);
  else
    hipLaunchKernelGGL((idx_adj_kernel<float>), gadj, dim3(256), 0, st, (const float*)R, mol_off, sq_off,
                       (float)cutoff, (float)int_cutoff, quad, w.adj, w.iadj);
  GN_LAUNCH_CHECK();
  const dim3 ga(gn_cdiv(A, 256));
  hipLaunchKernelGGL(idx_deg_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, quad, w.adj, w.iadj, w.deg,
                     w.up, w.ideg);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.up, w.off_half, (int64_t)A, w.overflow);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.deg, w.in_ptr, (int64_t)A, w.overflow);
  if (quad) hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.ideg, w.off_int, (int64_t)A, w.overflow);
  GN_LAUNCH_CHECK();
  int32_t tot[3] = {0, 0, 0};
  e = hipMemcpyAsync(&tot[0], w.off_half + A, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && quad) e = hipMemcpyAsync(&tot[1], w.off_int + A, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return (int)e;
  const int64_t H = tot[0], E = 2 * H, Eint = tot[1];
  sizes[0] = E;
  sizes[2] = E == 0 ? 0 : Eint;   // no-edge early return (data_container.py:282-285): every index array is empty
  if (E == 0) return 0;
  if (!id_a) return 0;    // size query only
  hipLaunchKernelGGL(idx_edges_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, quad, w.adj, w.iadj,
                     w.off_half, w.off_int, id_a, id_c, id_undir, id_swap, int_a, int_b, w.Mx, w.MI, 0x7fffffff, 0x7fffffff);
  hipLaunchKernelGGL(idx_in_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, w.adj, w.Mx, w.in_ptr,
                     w.in_edge, w.pos_in, (const int32_t*)nullptr, 0);
  hipLaunchKernelGGL(idx_cnt3_kernel, dim3(gn_cdiv(E, 256)), dim3(256), 0, st, id_a, w.deg, (int)E, w.cnt);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off3, E, w.overflow);
  GN_LAUNCH_CHECK();
  int32_t t3 = 0, t4 = 0, tca = 0, tdb = 0;
  int ovf = 0;
  e = hipMemcpyAsync(&t3, w.off3 + E, 4, hipMemcpyDeviceToHost, st);
  if (quad && Eint > 0) {
    hipLaunchKernelGGL(idx_cnt_intm_kernel, dim3(gn_cdiv(Eint, 256)), dim3(256), 0, st, int_a, int_b, w.deg, (int)Eint,
                       w.cnt, w.cnt2);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off_ca, Eint, w.overflow);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt2, w.off_db, Eint, w.overflow);
    hipLaunchKernelGGL(idx_cnt4_kernel, dim3(gn_cdiv(E, 256)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, id_a,
                       id_c, (int)E, w.adj, w.iadj, w.deg, w.cnt, (const int32_t*)nullptr, 0);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off4, E, w.overflow);
    GN_LAUNCH_CHECK();
    if (e == hipSuccess) e = hipMemcpyAsync(&tca, w.off_ca + Eint, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&tdb, w.off_db + Eint, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&t4, w.off4 + E, 4, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&ovf, w.overflow, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return (int)e;
  if (ovf) return (int)hipErrorInvalidValue;   // more than 2^31 - 1 entries in one array
  sizes[1] = t3;
  sizes[3] = tca;
  sizes[4] = tdb;
  sizes[5] = t4;
  return 0;
}

// Stage 2: fill the triplet / quadruplet arrays (sizes from stage 1; same `ws`, untouched in between).
extern "C" int gn_index_gpu_stage2(const int32_t* mol_off, const int32_t* sq_off, int B, int A, int64_t sum_n2,
                                   int triplets_only, void* ws, const int32_t* id_a, const int32_t* id_c,
                                   const int32_t* int_a, const int32_t* int_b, int64_t E, int64_t Eint,
                                   int32_t* id3_reduce_ca, int32_t* id3_expand_ba, int32_t* Kidx3,
                                   int32_t* id4_reduce_ca, int32_t* id4_expand_db, int32_t* id4_reduce_cab,
                                   int32_t* id4_expand_abd, int32_t* Kidx4, int32_t* id4_reduce_intm_ca,
                                   int32_t* id4_expand_intm_db, int32_t* id4_reduce_intm_ab,
                                   int32_t* id4_expand_intm_ab, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (E <= 0) return 0;
  const int quad = !triplets_only;
  idx_ws w;
  const int64_t emax = sum_n2 - A;
  idx_layout((char*)ws, A, sum_n2, emax, quad ? emax : 0, quad, &w);
  hipLaunchKernelGGL(idx_trip_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, id_a, id_c,
                     (int)E, w.adj, w.Mx, w.off3, id3_reduce_ca, id3_expand_ba, Kidx3, (const int32_t*)nullptr, 0, 0, (const int32_t*)nullptr);
  GN_LAUNCH_CHECK();
  if (quad && Eint > 0) {
    hipLaunchKernelGGL(idx_intm_kernel, dim3(gn_cdiv(Eint, 256)), dim3(256), 0, st, int_a, int_b, (int)Eint, w.in_ptr,
                       w.in_edge, w.off_ca, w.off_db, id4_reduce_intm_ca, id4_reduce_intm_ab, id4_expand_intm_db,
                       id4_expand_intm_ab, (const int32_t*)nullptr, (const int32_t*)nullptr);
    hipLaunchKernelGGL(idx_quad_kernel, dim3(gn_cdiv(E, 4)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, id_a, id_c,
                       (int)E, w.adj, w.iadj, w.Mx, w.MI, w.pos_in, w.off_ca, w.off_db, w.off4, id4_reduce_ca,
                       id4_expand_db, id4_reduce_cab, id4_expand_abd, Kidx4, (const int32_t*)nullptr, (const int32_t*)nullptr);
    GN_LAUNCH_CHECK();
  }
  return 0;
}

// Capacity form of the triplets-only build: no read-back, capturable.  See include/gemnet_hip.h.
extern "C" int gn_index_gpu_padded_t(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B, int A,
                                     int nmax, int64_t sum_n2, double cutoff, void* ws, int e_cap, int t_cap, int a_cap,
                                     int n_groups, int deg_bound, int32_t* staging, int32_t* id_c, int32_t* id_a,
                                     int32_t* id_swap, int32_t* id_undir, int32_t* id3_reduce_ca, int32_t* id3_expand_ba,
                                     int32_t* state, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (A <= 0 || B <= 0 || e_cap <= 0 || t_cap <= 0 || n_groups <= 0 || (e_cap & 1) || (t_cap & 1)) return (int)hipErrorInvalidValue;
  if (sum_n2 > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  idx_ws w;
  const int64_t emax = sum_n2 - A;
  idx_layout((char*)ws, A, sum_n2, emax, 0, 0, &w);
  const int n = (int)(emax < e_cap ? emax : e_cap);     // E <= emax always: rows the E-wide passes have to look at
  int32_t *s_a = staging, *s_c = staging + e_cap, *s_undir = staging + 2 * (size_t)e_cap, *s_swap = staging + 3 * (size_t)e_cap,
          *s_red = staging + 4 * (size_t)e_cap, *s_exp = staging + 4 * (size_t)e_cap + t_cap;
  hipError_t e = hipMemsetAsync(w.overflow, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(idx_atom_mol_kernel, dim3(B), dim3(64), 0, st, mol_off, B, w.atom_mol, (int32_t*)nullptr);
  dim3 gadj(gn_cdiv((int64_t)nmax * nmax, 256), B);
  if (r_is_f64)
    hipLaunchKernelGGL((idx_adj_kernel<double>), gadj, dim3(256), 0, st, (const double*)R, mol_off, sq_off, cutoff, cutoff, 0,
                       w.adj, w.iadj);
  else
    hipLaunchKernelGGL((idx_adj_kernel<float>), gadj, dim3(256), 0, st, (const float*)R, mol_off, sq_off, (float)cutoff,
                       (float)cutoff, 0, w.adj, w.iadj);
  const dim3 ga(gn_cdiv(A, 256));
  hipLaunchKernelGGL(idx_deg_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, 0, w.adj, w.iadj, w.deg, w.up, w.ideg);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.up, w.off_half, (int64_t)A, w.overflow);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.deg, w.in_ptr, (int64_t)A, w.overflow);
  GN_LAUNCH_CHECK();
  const int32_t* half_total = w.off_half + A;
  hipLaunchKernelGGL(idx_edges_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, 0, w.adj, w.iadj, w.off_half,
                     w.off_int, s_a, s_c, s_undir, s_swap, (int32_t*)nullptr, (int32_t*)nullptr, w.Mx, w.MI, e_cap, 0);
  hipLaunchKernelGGL(idx_in_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, w.adj, w.Mx, w.in_ptr, w.in_edge,
                     w.pos_in, half_total, e_cap);
  if (n > 0) {
    hipLaunchKernelGGL(idx_cnt3_cap_kernel, dim3(gn_cdiv(n, 256)), dim3(256), 0, st, s_a, w.deg, half_total, n, e_cap, w.cnt);
    hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off3, (int64_t)n, w.overflow);
    hipLaunchKernelGGL(idx_trip_kernel, dim3(gn_cdiv(n, 4)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, s_a, s_c, n, w.adj,
                       w.Mx, w.off3, s_red, s_exp, (int32_t*)nullptr, half_total, e_cap, t_cap, (const int32_t*)nullptr);
  } else {
    e = hipMemsetAsync(w.off3, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
  }
  GN_LAUNCH_CHECK();
  PadT p{s_c, s_a, s_swap, s_undir, s_red, s_exp, id_c, id_a, id_swap, id_undir, id3_reduce_ca, id3_expand_ba};
  const int64_t work = e_cap > t_cap ? e_cap : t_cap;
  hipLaunchKernelGGL(idx_commit_pad_t_kernel, dim3((unsigned)(gn_cdiv(work, 256) < 2048 ? gn_cdiv(work, 256) : 2048)), dim3(256), 0,
                     st, p, half_total, (const int32_t*)(w.off3 + n), e_cap, t_cap, a_cap, n_groups, deg_bound, state);
  GN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gn_index_poison_f32(float* x, int64_t n, const int32_t* state, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(idx_poison_kernel, dim3((unsigned)(gn_cdiv(n, 256) < 1024 ? gn_cdiv(n, 256) : 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, n, state);
  GN_LAUNCH_CHECK();
  return 0;
}

// Capacity form of the quadruplet build: no read-back, capturable.  See include/gemnet_hip.h.
extern "C" int gn_index_gpu_padded_q(const void* R, int r_is_f64, const int32_t* mol_off, const int32_t* sq_off, int B, int A,
                                     int nmax, int64_t sum_n2, double cutoff, double int_cutoff, void* ws, int e_cap, int t_cap,
                                     int eint_cap, int i_cap, int q_cap, int a_cap, int n_groups, int deg_bound,
                                     int32_t* staging, int32_t* const* arrays, int32_t* state, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (A <= 0 || B <= 0 || n_groups <= 0 || e_cap <= 0 || t_cap <= 0 || eint_cap <= 0 || i_cap <= 0 || q_cap <= 0 ||
      (e_cap & 1) || (t_cap & 1) || sum_n2 > 0x7fffffffLL)
    return (int)hipErrorInvalidValue;
  idx_ws w;
  const int64_t emax = sum_n2 - A;
  idx_layout((char*)ws, A, sum_n2, emax, emax, 1, &w);
  const int n = (int)(emax < e_cap ? emax : e_cap), ni = (int)(emax < eint_cap ? emax : eint_cap);
  int32_t *s_a = staging, *s_c = staging + e_cap, *s_undir = staging + 2 * (size_t)e_cap, *s_swap = staging + 3 * (size_t)e_cap,
          *s_int_a = staging + 4 * (size_t)e_cap, *s_int_b = staging + 4 * (size_t)e_cap + eint_cap;
  // arrays: id_c, id_a, id_swap, id_undir, id3_reduce_ca, id3_expand_ba, id4_int_a, id4_int_b, id4_reduce_intm_ca,
  //         id4_expand_intm_db, id4_reduce_intm_ab, id4_expand_intm_ab, id4_reduce_ca, id4_expand_db, id4_reduce_cab, id4_expand_abd
  hipError_t e = hipMemsetAsync(w.overflow, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(idx_atom_mol_kernel, dim3(B), dim3(64), 0, st, mol_off, B, w.atom_mol, (int32_t*)nullptr);
  dim3 gadj(gn_cdiv((int64_t)nmax * nmax, 256), B);
  if (r_is_f64)
    hipLaunchKernelGGL((idx_adj_kernel<double>), gadj, dim3(256), 0, st, (const double*)R, mol_off, sq_off, cutoff, int_cutoff,
                       1, w.adj, w.iadj);
  else
    hipLaunchKernelGGL((idx_adj_kernel<float>), gadj, dim3(256), 0, st, (const float*)R, mol_off, sq_off, (float)cutoff,
                       (float)int_cutoff, 1, w.adj, w.iadj);
  const dim3 ga(gn_cdiv(A, 256));
  hipLaunchKernelGGL(idx_deg_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, 1, w.adj, w.iadj, w.deg, w.up, w.ideg);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.up, w.off_half, (int64_t)A, w.overflow);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.deg, w.in_ptr, (int64_t)A, w.overflow);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.ideg, w.off_int, (int64_t)A, w.overflow);
  GN_LAUNCH_CHECK();
  const int32_t *half_total = w.off_half + A, *eint_total = w.off_int + A;
  hipLaunchKernelGGL(idx_edges_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, 1, w.adj, w.iadj, w.off_half,
                     w.off_int, s_a, s_c, s_undir, s_swap, s_int_a, s_int_b, w.Mx, w.MI, e_cap, eint_cap);
  hipLaunchKernelGGL(idx_in_kernel, ga, dim3(256), 0, st, mol_off, sq_off, w.atom_mol, A, w.adj, w.Mx, w.in_ptr, w.in_edge,
                     w.pos_in, half_total, e_cap);
  if (n <= 0 || ni <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(idx_cnt3_cap_kernel, dim3(gn_cdiv(n, 256)), dim3(256), 0, st, s_a, w.deg, half_total, n, e_cap, w.cnt);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off3, (int64_t)n, w.overflow);
  hipLaunchKernelGGL(idx_cnt_intm_cap_kernel, dim3(gn_cdiv(ni, 256)), dim3(256), 0, st, s_int_a, s_int_b, w.deg, eint_total, ni,
                     eint_cap, half_total, e_cap, w.cnt, w.cnt2);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off_ca, (int64_t)ni, w.overflow);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt2, w.off_db, (int64_t)ni, w.overflow);
  hipLaunchKernelGGL(idx_cnt4_kernel, dim3(gn_cdiv(n, 256)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, s_a, s_c, n, w.adj,
                     w.iadj, w.deg, w.cnt, half_total, e_cap);
  hipLaunchKernelGGL(idx_scan_kernel, dim3(1), dim3(1024), 0, st, w.cnt, w.off4, (int64_t)n, w.overflow);
  GN_LAUNCH_CHECK();
  const QCounts c{half_total, w.off3 + n, eint_total, w.off_ca + ni, w.off_db + ni, w.off4 + n};
  const QCaps k{e_cap, t_cap, eint_cap, i_cap, q_cap, a_cap, n_groups, deg_bound};
  hipLaunchKernelGGL(idx_decide_q_kernel, dim3(1), dim3(64), 0, st, c, k, state);
  // the writers of the large families: straight into the model's arrays, or nothing at all
  hipLaunchKernelGGL(idx_trip_kernel, dim3(gn_cdiv(n, 4)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, s_a, s_c, n, w.adj,
                     w.Mx, w.off3, arrays[4], arrays[5], (int32_t*)nullptr, half_total, e_cap, t_cap, (const int32_t*)state);
  hipLaunchKernelGGL(idx_intm_kernel, dim3(gn_cdiv(ni, 256)), dim3(256), 0, st, s_int_a, s_int_b, ni, w.in_ptr, w.in_edge,
                     w.off_ca, w.off_db, arrays[8], arrays[10], arrays[9], arrays[11], eint_total, (const int32_t*)state);
  hipLaunchKernelGGL(idx_quad_kernel, dim3(gn_cdiv(n, 4)), dim3(256), 0, st, mol_off, sq_off, w.atom_mol, s_a, s_c, n, w.adj,
                     w.iadj, w.Mx, w.MI, w.pos_in, w.off_ca, w.off_db, w.off4, arrays[12], arrays[13], arrays[14], arrays[15],
                     (int32_t*)nullptr, half_total, (const int32_t*)state);
  GN_LAUNCH_CHECK();
  PadQ p{s_c, s_a, s_swap, s_undir, s_int_a, s_int_b, arrays[0], arrays[1], arrays[2], arrays[3], arrays[4], arrays[5],
         arrays[6], arrays[7], arrays[8], arrays[9], arrays[10], arrays[11], arrays[12], arrays[13], arrays[14], arrays[15]};
  hipLaunchKernelGGL(idx_commit_pad_q_kernel, dim3(2048), dim3(256), 0, st, p, k, (const int32_t*)state);
  GN_LAUNCH_CHECK();
  return 0;
}
