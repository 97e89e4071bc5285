// LDS-resident layer chains, "wide" layout: 4 waves x 32 output columns, SEVERAL workgroups per CU
// (include/gemnet_hip.h, gn_chain_split_f32 with GN_CHAIN_WIDE; the two-plane fp16 arithmetic of chain2.hip "format H").
//
// Why a second layout.  chain2.hip runs ONE 8-wave workgroup per CU (an 80-row tile fills LDS and registers), every wave
// owning 16 output columns for all rows.  Measured there (tools/chain2_trace.py): a GEMM op of an
// 80-row tile takes 7.0 k cycles of which the matrix pipe is busy 1.9 k — the rest is exposed latency that a single
// workgroup in lockstep (one barrier per op) has nothing to overlap with: descriptor fetch 0.9 k, weight wait 0.6 k, the
// epilogue's global operands, 1.2 k of barrier skew, 4.2 k per LOAD op; and the MFMA phase itself is bound by LDS reads
// (every one of the 8 waves reads the whole row tile: 320 KB per op).  Here:
//   * a workgroup is 4 waves and a row tile of at most 48 rows, so TWO workgroups (each with its own barrier domain) share
//     a CU: one's epilogue / LOAD / weight wait runs under the other's MFMA phase;
//   * a wave owns 32 output columns (two 16-column tiles): an X fragment read from LDS feeds six MFMAs instead of three —
//     half the LDS read volume per row (4 x redundancy instead of 8 x);
//   * the tile height is a RUNTIME number of rows (a multiple of 8, at most 16 RT): 18 122 edge rows become 454 tiles of 40
//     rows = two per CU on 227 CUs — the same 80 rows per busy CU as before, in two independent halves.  (Tiles of a whole
//     number of 16-row MFMA blocks would leave 96 rows on the busiest CU.)
// Everything else — program format, LDS plane layout and swizzle, packed weight format, epilogue stages, row scales of linear
// programs, second outputs, source terms — is chain2.hip's; see there for the reasoning behind each of them.  NOT here: the
// register parking slot (slot 2): a wave holds twice the weights and accumulators of chain2.hip's and has no registers left
// for it — programs that use it take the 8-wave layout (gn_chain_split_f32 decides).
#include "common.h"

#include "chain_split.h"

namespace {
using namespace gn_split;

constexpr int NT3 = 256;     // 4 waves
constexpr int CT = 2;        // 16-column tiles per wave: tiles 2 wave, 2 wave + 1

template <int RT, bool ADJ>
__global__ __launch_bounds__(NT3, 2) void chain_wide_kernel(const gn_chain_args P, const int tile_rows, const int stagger) {
  constexpr bool HF = true;
  constexpr int NPL = 2;
  constexpr int BM = 16 * RT;
  constexpr int PLANE = BM * ROWB;
  constexpr int SP = 2;
  constexpr int SLOT = SP * PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int l15 = lane & 15;
  const int lg = lane >> 4;
  const int M = P.M;
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;
  // rows of this tile that exist: everything read from or written to global memory is limited to r < rlim
  const int rlim = (int)(((int64_t)M - row0) < (int64_t)tile_rows ? ((int64_t)M - row0) : (int64_t)tile_rows);
  __shared__ const void* gemm_W[GN_CHAIN_MAX_OPS + 2];
  __shared__ int gemm_NK[GN_CHAIN_MAX_OPS + 2];
  __shared__ float rs[2][16 * 3];
  __shared__ int prog_linear;
  if (tid == 0) {
    int g = 0, lin = 1;
    for (int j = 0; j < P.n_ops; ++j)
      if (P.ops[j].kind == GN_OP_GEMM) {
        gemm_W[g] = P.ops[j].W;
        gemm_NK[g++] = (P.ops[j].N << 16) | P.ops[j].K;
        if (P.ops[j].act & 1) lin = 0;
      }
    gemm_W[g] = gemm_W[g + 1] = nullptr;
    gemm_NK[g] = gemm_NK[g + 1] = 0;
    prog_linear = lin;
  }
  if (tid < 2 * 16 * 3) (&rs[0][0])[tid] = 1.f;
  __syncthreads();
  // Tuning hook (GN_CHAIN_WIDE_STAGGER bits of `nprod`, default 0): the workgroup whose LDS allocation does not start at 0 — the second to
  // arrive on its CU — starts `stagger` x 64 cycles late, so that one's MFMA phase meets the other's epilogue.  Measured
  // (profiles/r4_chain_layouts.txt): a pair offset by one op finishes exactly that much later — the two workgroups do not
  // disturb each other in phase either; per-wave latency, not contention, sets the time of an op.
  if (stagger > 0 && (__builtin_amdgcn_s_getreg((31 << 11) | 6) & 0xff) != 0) {      // HW_REG_LDS_ALLOC: LDS_BASE
    for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(1);
  }
  const bool scaled = prog_linear != 0;   // uniform: linear programs carry a power-of-two scale per row (chain2.hip)
  float sg[2][RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) sg[0][t] = sg[1][t] = 1.f;
  auto sg_get = [&](int slot, int t) -> float { return slot ? sg[1][t] : sg[0][t]; };
  auto sg_set = [&](int slot, int t, float v) { if (slot) sg[1][t] = v; else sg[0][t] = v; };

  // weight fragments of one GEMM op: [column tile][k-chunk][plane]
  uint4 bcur[CT][4][NPL];
  auto wload_op = [&](int ord) {
    const int nk = __builtin_amdgcn_readfirstlane(gemm_NK[ord]);
    const int N = nk >> 16, kc = ((nk & 0xffff) + 31) >> 5;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int tile = 2 * wave + ct;
      const uint4* __restrict__ base = reinterpret_cast<const uint4*>(gemm_W[ord]) + ((size_t)tile * kc * SP) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          bcur[ct][c][p] = make_uint4(0u, 0u, 0u, 0u);
          if (tile * 16 < N && c < kc) bcur[ct][c][p] = base[(c * SP + p) * 64];
        }
    }
  };
  wload_op(0);
  int gord = 0;

  auto plane_ptr = [&](int slot, int p) -> unsigned char* { return smem + slot * SLOT + p * PLANE; };
  // per-wave staging area of the epilogue factor: [column tile][row block] KB
  unsigned char* const stage = smem + 2 * SLOT + wave * (CT * RT * 1024);
  const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;
  auto sw_off = [&](int row, int col) -> int { return row * ROWB + ((((col >> 3) ^ row) & 15) << 4) + ((col & 4) << 1); };
  auto slot_write = [&](int slot, int row, int col, const float4 v) {
    const int off = sw_off(row, col);
    uint2 H, L;
    split4h(v, H, L);
    *reinterpret_cast<uint2*>(plane_ptr(slot, 0) + off) = H;
    *reinterpret_cast<uint2*>(plane_ptr(slot, 1) + off) = L;
  };
  auto slot_read = [&](int slot, int row, int col) -> float4 {
    const int off = sw_off(row, col);
    return join4h(*reinterpret_cast<const uint2*>(plane_ptr(slot, 0) + off),
                  *reinterpret_cast<const uint2*>(plane_ptr(slot, 1) + off));
  };
  auto mul4 = [](float4 v, float a) -> float4 { return make_float4(v.x * a, v.y * a, v.z * a, v.w * a); };
  // accumulator-layout read of an LDS slot in TRUE scale: row 16 t + l15, columns 16 tile + 4 lg .. +3
  auto slot_read_acc = [&](int slot, int t, int tile) -> float4 {
    const float4 v = slot_read(slot, 16 * t + l15, tile * 16 + (lg << 2));
    return mul4(v, inv_pow2(sg_get(slot, t)));
  };

  for (int oi = 0; oi < P.n_ops; ++oi) {
    const gn_chain_op op = P.ops[oi];
    const int kind = op.kind;
    if (kind == GN_OP_LOAD) {
      const int width = op.width, slot = op.slot, ld = op.ld, y2_slot = op.y2_slot, mode2 = op.mode2;
      const int w4 = ((width + 31) & ~31) >> 2;
      const float alpha = op.alpha, alpha2 = op.alpha2;
      const float* __restrict__ const src = op.src;
      const float* __restrict__ const Z2 = op.Z2;
      const int32_t* __restrict__ const rows = op.rows;
      const bool lsrc = ADJ && op.src_stage == 2 && y2_slot >= 0 && op.srcP;
      const float* __restrict__ const srcP = op.srcP;
      const float* __restrict__ const srcQ = op.srcQ;
      const int src_mode = op.src_mode;
      const float src_alpha = op.src_alpha;
      // a [BM][<= 32 float4] tile is 2 RT passes of the 256 threads, taken in two halves of RT passes (all loads of a half in
      // flight before its first use; both halves at once cost 24 more registers than the GEMM ops leave)
      constexpr int LI = RT;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
      float4 v[LI], zz[LI];
      bool in[LI];
#pragma unroll
      for (int i = 0; i < LI; ++i) {
        const int f = tid + (half * RT + i) * NT3;
        const int r = f / w4, c = (f - r * w4) << 2;
        const int64_t gr = row0 + r;
        in[i] = f < BM * w4 && r < rlim && c < width;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        zz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[i]) {
          const int64_t sr = rows ? (int64_t)rows[gr] : gr;
          v[i] = *reinterpret_cast<const float4*>(src + sr * ld + c);
          if (ADJ && y2_slot >= 0 && Z2) zz[i] = *reinterpret_cast<const float4*>(Z2 + gr * width + c);
        }
      }
      float sig[LI];
#pragma unroll
      for (int i = 0; i < LI; ++i) {
        v[i].x *= alpha; v[i].y *= alpha; v[i].z *= alpha; v[i].w *= alpha;
        sig[i] = 1.f;
      }
      if (scaled && (w4 == 8 || w4 == 16 || w4 == 32)) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
          const float m = fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
          sig[i] = row_sigma(group_max(m, w4));
        }
      }
#pragma unroll
      for (int i = 0; i < LI; ++i) {
        const int f = tid + (half * RT + i) * NT3;
        if (f < BM * w4) {
          const int r = f / w4, c = (f - r * w4) << 2;
          slot_write(slot, r, c, mul4(v[i], sig[i]));
          if (c == 0) { rs[slot][r] = sig[i]; if (ADJ && y2_slot >= 0) rs[y2_slot][r] = sig[i]; }
          if (ADJ && y2_slot >= 0) {
            float4 u = make_float4(v[i].x * alpha2, v[i].y * alpha2, v[i].z * alpha2, v[i].w * alpha2);
            if (Z2 && in[i]) {
              const float4 z = zz[i];
              if (mode2 == 0) { u.x *= gn_dssilu(z.x); u.y *= gn_dssilu(z.y); u.z *= gn_dssilu(z.z); u.w *= gn_dssilu(z.w); }
              else if (mode2 == 1) { u.x *= z.x; u.y *= z.y; u.z *= z.z; u.w *= z.w; }
              else { u.x *= gn_ssilu(z.x); u.y *= gn_ssilu(z.y); u.z *= gn_ssilu(z.z); u.w *= gn_ssilu(z.w); }
            }
            if (lsrc && in[i]) {
              const int64_t gr = row0 + r;
              const float4 p = *reinterpret_cast<const float4*>(srcP + gr * width + c);
              const float4 q = srcQ ? *reinterpret_cast<const float4*>(srcQ + gr * width + c) : make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 sv = src_term(zz[i], p, q, src_mode, src_alpha);
              u.x += sv.x; u.y += sv.y; u.z += sv.z; u.w += sv.w;
            }
            slot_write(y2_slot, r, c, mul4(u, sig[i]));
          }
        }
      }
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        sg_set(slot, t, rs[slot][16 * t + l15]);
        if (ADJ && y2_slot >= 0) sg_set(y2_slot, t, rs[y2_slot][16 * t + l15]);
      }
    } else if (kind == GN_OP_SCALE) {
      const int slot = op.slot, a_slot = op.a_slot, ld = op.ld, width = op.width;
      const float alpha = op.alpha;
      const int mode = op.act;
      const float* __restrict__ const src = op.src;
      float* __restrict__ const out = op.out;
      {
        const int w4 = width >> 2;
        for (int f = tid; f < BM * w4; f += NT3) {
          const int r = f / w4, c = (f - r * w4) << 2;
          const int64_t gr = row0 + r;
          const bool live = r < rlim;
          const float sc = rs[a_slot][r];
          float4 v = slot_read(a_slot, r, c);
          const float a_true = alpha * inv_pow2(sc);
          v.x *= a_true; v.y *= a_true; v.z *= a_true; v.w *= a_true;
          float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          if (src && live) {
            z = *reinterpret_cast<const float4*>(src + gr * ld + c);
            if (mode == 0) { v.x *= gn_dssilu(z.x); v.y *= gn_dssilu(z.y); v.z *= gn_dssilu(z.z); v.w *= gn_dssilu(z.w); }
            else if (mode == 1) { v.x *= z.x; v.y *= z.y; v.z *= z.z; v.w *= z.w; }
            else { v.x *= gn_ssilu(z.x); v.y *= gn_ssilu(z.y); v.z *= gn_ssilu(z.z); v.w *= gn_ssilu(z.w); }
          }
          if (ADJ && op.src_stage == 1 && op.srcP && live) {
            const float4 p = *reinterpret_cast<const float4*>(op.srcP + gr * ld + c);
            const float4 q = op.srcQ ? *reinterpret_cast<const float4*>(op.srcQ + gr * ld + c) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sv = src_term(z, p, q, op.src_mode, op.src_alpha);
            v.x += sv.x; v.y += sv.y; v.z += sv.z; v.w += sv.w;
          }
          slot_write(slot, r, c, mul4(v, sc));
          if (c == 0 && slot != a_slot) rs[slot][r] = sc;
          if (out && live) *reinterpret_cast<float4*>(out + gr * ld + c) = v;
        }
        lds_barrier();
#pragma unroll
        for (int t = 0; t < RT; ++t) sg_set(slot, t, sg_get(a_slot, t));
      }
    } else if (kind == GN_OP_STORE) {
      const int w4 = op.width >> 2, slot = op.slot, ld = op.ld;
      float* __restrict__ const out = op.out;
      for (int f = tid; f < BM * w4; f += NT3) {
        const int r = f / w4, c = (f - r * w4) << 2;
        if (r < rlim) {
          const float4 v = slot_read(slot, r, c);
          *reinterpret_cast<float4*>(out + (row0 + r) * ld + c) = mul4(v, inv_pow2(rs[slot][r]));
        }
      }
      lds_barrier();
    } else {  // GN_OP_GEMM
      const int N = op.N, K = op.K, a_slot = op.a_slot, y_slot = op.slot, act = op.act & 1;
      const bool pre_deriv = (op.act & 2) != 0;
      const float alpha = op.alpha, beta = op.beta, beta2 = op.beta2;
      const float* __restrict__ const gadd1 = op.gadd1;
      const float* __restrict__ const gadd2 = op.gadd2;
      const int32_t* __restrict__ const gidx1 = op.gidx1;
      const int32_t* __restrict__ const gidx2 = op.gidx2;
      float* __restrict__ const pre_out = op.pre_out;
      float* __restrict__ const out = op.out;
      const int mul_slot = op.mul_slot, res_slot = op.res_slot, res2_slot = op.res2_slot;
      const float* __restrict__ const mul_g = op.mul_g;
      const float* __restrict__ const res_g = op.res_g;
      const float* __restrict__ const res2_g = op.res2_g;
      const int32_t* __restrict__ const res_rows = op.res_rows;
      const int mul_mode = op.mul_mode, y2_slot = op.y2_slot, y2_src = op.y2_src, mode2 = op.mode2;
      const float alpha2 = op.alpha2;
      const float* __restrict__ const Z2 = op.Z2;
      float* __restrict__ const out2 = op.out2;
      const int src_stage = ADJ ? op.src_stage : 0, src_mode = op.src_mode;
      const float src_alpha = op.src_alpha;
      const float* __restrict__ const srcP = op.srcP;
      const float* __restrict__ const srcQ = op.srcQ;
      const bool act0 = (2 * wave) * 16 < N, act1 = (2 * wave + 1) * 16 < N;     // wave-uniform
      ++gord;

      bool ok[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) ok[t] = 16 * t + l15 < rlim;

      const bool staged = mul_g != nullptr && mul_slot < 0;
      if (staged) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int n0 = (2 * wave + ct) * 16 + (lg << 2);
          if (ct == 0 ? act0 : act1) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
              if (ok[t]) g2lds16(mul_g + (uint32_t)(row0 + 16 * t + l15) * (uint32_t)N + (uint32_t)n0, stage_lds + (ct * RT + t) * 1024);
          }
        }
      }
      // per column tile: hh | hl | lh (the two correction terms in registers of their own, as in chain2.hip: no back-to-back
      // dependent MFMAs, and the same summation order — the layouts stay bit-identical)
      v4f acc[CT][3][RT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[ct][0][t] = acc[ct][1][t] = acc[ct][2][t] = (v4f){0.f, 0.f, 0.f, 0.f};
      if (act0) {
        const unsigned char* xb = smem + a_slot * SLOT + l15 * ROWB;
        const int kc = (K + 31) >> 5;
        // X fragments one step ahead (chain2.hip) — except at RT = 3, where the second buffer is the 8 registers the kernel
        // does not have (it spilled them): there the co-resident workgroup covers the LDS latency
        constexpr bool PF = RT < 3;
        uint4 xf[PF ? 2 : 1][2];
        auto xload = [&](uint4 (&f)[2], int c, int t) {
          const unsigned char* xp = xb + (16 * t) * ROWB + ((((c << 2) | lg) ^ l15) << 4);
          f[0] = *reinterpret_cast<const uint4*>(xp);
          f[1] = *reinterpret_cast<const uint4*>(xp + PLANE);
        };
        if (PF) xload(xf[0], 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < kc) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
              const int cur = PF ? ((c * RT + t) & 1) : 0;
              if (PF) {
                const int s2 = c * RT + t + 1, c2 = s2 / RT, t2 = s2 - c2 * RT;   // the next step's fragments
                if (c2 < 4 && c2 < kc) xload(xf[PF ? (s2 & 1) : 0], c2, t2);
              } else {
                xload(xf[0], c, t);
              }
              const f16x8 yh = __builtin_bit_cast(f16x8, xf[cur][0]), yl = __builtin_bit_cast(f16x8, xf[cur][1]);
              {
                const f16x8 ah = __builtin_bit_cast(f16x8, bcur[0][c][0]), al = __builtin_bit_cast(f16x8, bcur[0][c][1]);
                acc[0][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, acc[0][0][t], 0, 0, 0);
                acc[0][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, acc[0][1][t], 0, 0, 0);
                acc[0][2][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, acc[0][2][t], 0, 0, 0);
              }
              if (act1) {
                const f16x8 ah = __builtin_bit_cast(f16x8, bcur[1][c][0]), al = __builtin_bit_cast(f16x8, bcur[1][c][1]);
                acc[1][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yh, acc[1][0][t], 0, 0, 0);
                acc[1][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, yl, acc[1][1][t], 0, 0, 0);
                acc[1][2][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, yh, acc[1][2][t], 0, 0, 0);
              }
            }
          }
        }
      }
      float4 v[CT][RT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const v4f s = acc[ct][0][t] + (acc[ct][1][t] + acc[ct][2][t]) * H_DOWN;
          v[ct][t] = make_float4(s[0], s[1], s[2], s[3]);
        }
      // the products are sigma_a times the true values; what this op leaves in LDS takes the smallest scale among its LDS operands
      float sy[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float sa = sg_get(a_slot, t);
        const float inv = inv_pow2(sa);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) v[ct][t] = mul4(v[ct][t], inv);
        float m = sa;
        if (res_slot == 0 || res_slot == 1) m = fminf(m, sg_get(res_slot, t));
        if (res2_slot == 0 || res2_slot == 1) m = fminf(m, sg_get(res2_slot, t));
        sy[t] = m;
      }
      wload_op(gord);   // next GEMM's fragments: in flight under the epilogue
      if (y_slot == a_slot || y2_slot == a_slot) lds_barrier();   // all reads of a_slot must finish before it is overwritten

#define GN3_EACH(body) _Pragma("unroll") for (int t = 0; t < RT; ++t) { body }
#define GN3_V v[ct][t]
#define GN3_ADD(q) GN3_V.x += q.x; GN3_V.y += q.y; GN3_V.z += q.z; GN3_V.w += q.w;
#define GN3_MUL(q) GN3_V.x *= q.x; GN3_V.y *= q.y; GN3_V.z *= q.z; GN3_V.w *= q.w;
#define GN3_RES(q, b) GN3_V.x = (GN3_V.x + q.x) * b; GN3_V.y = (GN3_V.y + q.y) * b; GN3_V.z = (GN3_V.z + q.z) * b; GN3_V.w = (GN3_V.w + q.w) * b;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int tile = 2 * wave + ct;
        if (!(ct == 0 ? act0 : act1)) continue;
        const int n0 = tile * 16 + (lg << 2);
        uint32_t off[RT];        // element offset of (row, n0) in an (M, N) matrix: M * 128 < 2^32 (checked on the host)
        GN3_EACH(off[t] = (uint32_t)(row0 + 16 * t + l15) * (uint32_t)N + (uint32_t)n0;)
        if (gadd1) {
          float4 q[RT];
          GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                   if (ok[t]) q[t] = *reinterpret_cast<const float4*>(gadd1 + (size_t)gidx1[row0 + 16 * t + l15] * N + n0);)
          GN3_EACH(GN3_ADD(q[t]))
        }
        if (gadd2) {
          float4 q[RT];
          GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                   if (ok[t]) q[t] = *reinterpret_cast<const float4*>(gadd2 + (size_t)gidx2[row0 + 16 * t + l15] * N + n0);)
          GN3_EACH(GN3_ADD(q[t]))
        }
        if (pre_out && !pre_deriv) GN3_EACH(if (ok[t]) *reinterpret_cast<float4*>(pre_out + off[t]) = GN3_V;)
        if (act && pre_out && pre_deriv) {
          GN3_EACH(
            float4 d;
            gn_ssilu_pair(GN3_V.x, GN3_V.x, d.x); gn_ssilu_pair(GN3_V.y, GN3_V.y, d.y);
            gn_ssilu_pair(GN3_V.z, GN3_V.z, d.z); gn_ssilu_pair(GN3_V.w, GN3_V.w, d.w);
            if (ok[t]) *reinterpret_cast<float4*>(pre_out + off[t]) = d;)
        } else if (act) GN3_EACH(GN3_V.x = gn_ssilu(GN3_V.x); GN3_V.y = gn_ssilu(GN3_V.y); GN3_V.z = gn_ssilu(GN3_V.z); GN3_V.w = gn_ssilu(GN3_V.w);)
        const bool want2 = ADJ && (y2_slot >= 0 || out2);
        auto emit_y2 = [&]() {
          float4 q[RT];
          if (Z2) {
            GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(Z2 + off[t]);)
            if (mode2 == 0) GN3_EACH(q[t] = make_float4(gn_dssilu(q[t].x), gn_dssilu(q[t].y), gn_dssilu(q[t].z), gn_dssilu(q[t].w));)
            else if (mode2 == 2) GN3_EACH(q[t] = make_float4(gn_ssilu(q[t].x), gn_ssilu(q[t].y), gn_ssilu(q[t].z), gn_ssilu(q[t].w));)
            GN3_EACH(q[t].x *= GN3_V.x * alpha2; q[t].y *= GN3_V.y * alpha2; q[t].z *= GN3_V.z * alpha2; q[t].w *= GN3_V.w * alpha2;)
          } else {
            GN3_EACH(q[t] = make_float4(GN3_V.x * alpha2, GN3_V.y * alpha2, GN3_V.z * alpha2, GN3_V.w * alpha2);)
          }
          if (src_stage == 2 && srcP) {
            GN3_EACH(if (ok[t]) {
              const float4 zs = Z2 ? *reinterpret_cast<const float4*>(Z2 + off[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
              const float4 pp = *reinterpret_cast<const float4*>(srcP + off[t]);
              const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off[t]) : make_float4(1.f, 1.f, 1.f, 1.f);
              const float4 sv = src_term(zs, pp, qq, src_mode, src_alpha);
              q[t].x += sv.x; q[t].y += sv.y; q[t].z += sv.z; q[t].w += sv.w;
            })
          }
          if (out2) GN3_EACH(if (ok[t]) *reinterpret_cast<float4*>(out2 + off[t]) = q[t];)
          if (y2_slot >= 0) GN3_EACH(slot_write(y2_slot, 16 * t + l15, n0, ok[t] ? mul4(q[t], sy[t]) : make_float4(0.f, 0.f, 0.f, 0.f));)
        };
        if (want2 && y2_src) emit_y2();
        if (mul_slot >= 0) GN3_EACH(const float4 q = slot_read_acc(mul_slot, t, tile); GN3_MUL(q))
        else if (mul_g) {
          float4 q[RT];
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS writes of the staged loads have landed
          GN3_EACH(q[t] = make_float4(1.f, 1.f, 1.f, 1.f);
                   if (ok[t]) q[t] = *reinterpret_cast<const float4*>(stage + (ct * RT + t) * 1024 + lane * 16);)
          if (mul_mode == 2) GN3_EACH(q[t] = make_float4(gn_dssilu(q[t].x), gn_dssilu(q[t].y), gn_dssilu(q[t].z), gn_dssilu(q[t].w));)
          else if (mul_mode == 3) GN3_EACH(q[t] = make_float4(gn_ssilu(q[t].x), gn_ssilu(q[t].y), gn_ssilu(q[t].z), gn_ssilu(q[t].w));)
          GN3_EACH(GN3_MUL(q[t]))
        }
        if (alpha != 1.0f) GN3_EACH(GN3_V.x *= alpha; GN3_V.y *= alpha; GN3_V.z *= alpha; GN3_V.w *= alpha;)
        if (src_stage == 1 && srcP) {
          GN3_EACH(if (ok[t]) {
            const float4 zs = (mul_g && src_mode == 1) ? *reinterpret_cast<const float4*>(mul_g + off[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 pp = *reinterpret_cast<const float4*>(srcP + off[t]);
            const float4 qq = srcQ ? *reinterpret_cast<const float4*>(srcQ + off[t]) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sv = src_term(zs, pp, qq, src_mode, src_alpha);
            GN3_V.x += sv.x; GN3_V.y += sv.y; GN3_V.z += sv.z; GN3_V.w += sv.w;
          })
        }
        if (res_slot >= 0) GN3_EACH(const float4 q = slot_read_acc(res_slot, t, tile); GN3_RES(q, beta))
        else if (res_g) {
          float4 q[RT];
          if (res_rows) GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                                 if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res_g + (size_t)res_rows[row0 + 16 * t + l15] * N + n0);)
          else GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res_g + off[t]);)
          GN3_EACH(if (ok[t]) { GN3_RES(q[t], beta) })
        }
        if (res2_slot >= 0) GN3_EACH(const float4 q = slot_read_acc(res2_slot, t, tile); GN3_RES(q, beta2))
        else if (res2_g) {
          float4 q[RT];
          GN3_EACH(q[t] = make_float4(0.f, 0.f, 0.f, 0.f); if (ok[t]) q[t] = *reinterpret_cast<const float4*>(res2_g + off[t]);)
          GN3_EACH(if (ok[t]) { GN3_RES(q[t], beta2) })
        }
        if (out) GN3_EACH(if (ok[t]) *reinterpret_cast<float4*>(out + off[t]) = GN3_V;)
        if (y_slot >= 0) GN3_EACH(slot_write(y_slot, 16 * t + l15, n0, ok[t] ? mul4(GN3_V, sy[t]) : make_float4(0.f, 0.f, 0.f, 0.f));)
        if (want2 && !y2_src) emit_y2();
        __builtin_amdgcn_sched_barrier(0);   // keep the two column tiles' epilogues apart (interleaved, they spill)
      }
#undef GN3_EACH
#undef GN3_V
#undef GN3_ADD
#undef GN3_MUL
#undef GN3_RES
      if ((N & 16) && y_slot >= 0 && y_slot < 2 && wave == (N >> 5)) {
        // N = 16 (mod 32): the next GEMM reads k-chunks of 32 columns, so the 16 columns after N are zeroed — column tile
        // N / 16 (odd: the second tile of wave N / 32, inactive above)
        const int n0 = N + (lg << 2);
#pragma unroll
        for (int t = 0; t < RT; ++t) slot_write(y_slot, 16 * t + l15, n0, make_float4(0.f, 0.f, 0.f, 0.f));
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float m = sy[t];
        if (y_slot == 0 || y_slot == 1) { sg_set(y_slot, t, m); if (wave == 0 && lg == 0) rs[y_slot][16 * t + l15] = m; }
        if (ADJ && y2_slot >= 0) { sg_set(y2_slot, t, m); if (wave == 0 && lg == 0) rs[y2_slot][16 * t + l15] = m; }
      }
      lds_barrier();
    }
  }
}

// stagger: start-up offset of the second workgroup of a CU, in units of 64 cycles (GN_CHAIN_WIDE_STAGGER bits of `nprod`)
template <int RT, bool ADJ>
int launch_chain_wide(const gn_chain_args* args, int tile_rows, int stagger, hipStream_t st) {
  constexpr int BM = 16 * RT;
  constexpr size_t smem = (size_t)2 * 2 * BM * ROWB + (size_t)4 * CT * RT * 1024;
  static std::atomic<bool> configured{false};   // set-once flag of an idempotent attribute (two racing threads both set it)
  if (!configured.load(std::memory_order_acquire)) {
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_wide_kernel<RT, ADJ>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    configured.store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((chain_wide_kernel<RT, ADJ>), dim3(gn_cdiv(args->M, tile_rows)), dim3(NT3), smem, st, *args, tile_rows,
                     stagger);
  GN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Rows per tile of the wide layout: one round of (at most) two workgroups per CU when the rows allow it, else rounds of
// 48-row tiles balanced over the rounds; the atom-side programs (M <= 4 k rows) keep one 16-row MFMA block per workgroup.
// GN_CHAIN_WIDE_ROWS(r) in the `nprod` argument of gn_chain_split_f32, r a multiple of 8 in 8..48, fixes the height for that
// launch (tuning and tests); 0 = this automatic choice.
extern "C" int gn_chain_wide_tile_rows(int M) {
  if (M <= 16 * 256) return 16;
  const int slots = 512;
  int per = gn_cdiv(M, slots);
  if (per > 48) {
    const int rounds = gn_cdiv(M, (int64_t)48 * slots);
    per = gn_cdiv(M, (int64_t)slots * rounds);
  }
  per = (per + 7) / 8 * 8;
  return per < 16 ? 16 : (per > 48 ? 48 : per);
}

// called by gn_chain_split_f32 (chain2.hip) for nprod = GN_CHAIN_F16X2 | GN_CHAIN_WIDE, after its argument checks
int gn_chain_wide_dispatch(const gn_chain_args* args, bool adj, int forced_tile_rows, int stagger, hipStream_t st) {
  const int tr = forced_tile_rows ? forced_tile_rows : gn_chain_wide_tile_rows(args->M);
  const int rt = gn_cdiv(tr, 16);
  if (adj) {
    if (rt == 1) return launch_chain_wide<1, true>(args, tr, stagger, st);
    if (rt == 2) return launch_chain_wide<2, true>(args, tr, stagger, st);
    return launch_chain_wide<3, true>(args, tr, stagger, st);
  }
  if (rt == 1) return launch_chain_wide<1, false>(args, tr, stagger, st);
  if (rt == 2) return launch_chain_wide<2, false>(args, tr, stagger, st);
  return launch_chain_wide<3, false>(args, tr, stagger, st);
}
