// Which wave layout should the split-operand chain kernel (csrc/chain2.hip) use?  One GEMM op of an 80-row tile
// (K = N = 128, six bf16 products) in a loop, operands resident: X planes in LDS (swizzled as in chain2.hip), W
// fragments in registers, epilogue = 3-plane split + ds_write_b64 into the other slot + LDS barrier.  No global traffic:
// isolates LDS reads + MFMA + epilogue, the three things the layout changes.
//   A  8 waves, wave = 16 columns x 5 row blocks                 (today: every wave reads the whole X tile, 480 KB / op)
//   B  4 waves, wave = 32 columns x 5 row blocks, 96 W registers  (240 KB / op; one wave per SIMD, 512-register budget)
//   C  8 waves, wave = 32 columns x {3 | 2} row blocks            (240 KB / op; needs W streamed or shared in the product)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/chain_layout_bench.hip -o tools/exp/bin/chain_layout_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int RT = 5, BM = 16 * RT, ROWB = 256, PLANE = BM * ROWB, SLOT = 3 * PLANE;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void split4(const float4 x, uint2& H, uint2& M, uint2& L) {
  H.x = pk_bf16(x.x, x.y); H.y = pk_bf16(x.z, x.w);
  const float r0 = x.x - bf_lo(H.x), r1 = x.y - bf_hi(H.x), r2 = x.z - bf_lo(H.y), r3 = x.w - bf_hi(H.y);
  M.x = pk_bf16(r0, r1); M.y = pk_bf16(r2, r3);
  L.x = pk_bf16(r0 - bf_lo(M.x), r1 - bf_hi(M.x)); L.y = pk_bf16(r2 - bf_lo(M.y), r3 - bf_hi(M.y));
}
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ int sw_off(int row, int col) {
  return row * ROWB + ((((col >> 3) ^ row) & 15) << 4) + ((col & 4) << 1);
}

// NW waves; each wave owns CT column tiles (16 columns each) and the row blocks t with (t % RS) == its row share index
template <int NW, int CT, int RS>
__global__ __launch_bounds__(64 * NW) void op_loop(unsigned long long* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  constexpr int NCG = 8 / CT;                   // column groups per tile
  const int cg = wave % NCG, rs = wave / NCG;   // column group, row share
  constexpr int NT = (RT + RS - 1) / RS;        // row blocks of a wave (max)
  for (int i = tid; i < 2 * SLOT / 4; i += 64 * NW) reinterpret_cast<uint32_t*>(smem)[i] = 0x3f803f80u;   // bf16 1.0 pairs
  __syncthreads();
  uint4 w[CT][4][3];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t v = pk_bf16(seed * (1 + j + c + p + lane % 7) * 1e-3f, seed * 2e-3f);
        w[j][c][p] = make_uint4(v, v ^ 0x00010001u, v + 0x00020002u, v ^ 0x00030003u);
      }
  int cur = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    v4f a0[NT][CT], a1[NT][CT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int j = 0; j < CT; ++j) { a0[u][j] = (v4f){0, 0, 0, 0}; a1[u][j] = (v4f){0, 0, 0, 0}; }
    const unsigned char* xb = smem + cur * SLOT + l15 * ROWB;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int t = rs + u * RS;
        if (t < RT) {
          const unsigned char* xp = xb + (16 * t) * ROWB + ((((c << 2) | lg) ^ l15) << 4);
          const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp));
          const bf16x8 xm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + PLANE));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + 2 * PLANE));
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, w[j][c][0]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, w[j][c][1]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, w[j][c][2]);
            a1[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, a1[u][j], 0, 0, 0);
            a0[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, a0[u][j], 0, 0, 0);
            a1[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, a1[u][j], 0, 0, 0);
            a1[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, a1[u][j], 0, 0, 0);
            a1[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, a1[u][j], 0, 0, 0);
            a1[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, a1[u][j], 0, 0, 0);
          }
        }
      }
    }
    // epilogue: scale, split into planes, write the other slot (transposed tile: 4 consecutive columns per lane)
    unsigned char* yb = smem + (cur ^ 1) * SLOT;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int t = rs + u * RS;
      if (t < RT) {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const v4f s = a0[u][j] + a1[u][j];
          const float4 v = make_float4(s[0] * 1e-3f + 1.f, s[1] * 1e-3f + 1.f, s[2] * 1e-3f + 1.f, s[3] * 1e-3f + 1.f);
          uint2 H, M, L;
          split4(v, H, M, L);
          const int off = sw_off(16 * t + l15, (cg * CT + j) * 16 + (lg << 2));
          *reinterpret_cast<uint2*>(yb + off) = H;
          *reinterpret_cast<uint2*>(yb + PLANE + off) = M;
          *reinterpret_cast<uint2*>(yb + 2 * PLANE + off) = L;
        }
      }
    }
    lds_barrier();
    cur ^= 1;
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (tid == 0 && smem[cur * SLOT] == 0x77) out[0] = 0;   // keep the LDS traffic alive
}

// Layout A with the MFMAs of TWO row blocks interleaved (a dependent accumulation chain of the cross terms alternates
// between two accumulators: the next MFMA never waits for the result of the previous one)
__global__ __launch_bounds__(512) void op_loop_pairs(unsigned long long* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 2 * SLOT / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3f803f80u;
  __syncthreads();
  uint4 w[4][3];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const uint32_t v = pk_bf16(seed * (1 + c + p + lane % 7) * 1e-3f, seed * 2e-3f);
      w[c][p] = make_uint4(v, v ^ 0x00010001u, v + 0x00020002u, v ^ 0x00030003u);
    }
  int cur = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    v4f a0[RT], a1[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { a0[t] = (v4f){0, 0, 0, 0}; a1[t] = (v4f){0, 0, 0, 0}; }
    const unsigned char* xb = smem + cur * SLOT + l15 * ROWB;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 wh = __builtin_bit_cast(bf16x8, w[c][0]);
      const bf16x8 wm = __builtin_bit_cast(bf16x8, w[c][1]);
      const bf16x8 wl = __builtin_bit_cast(bf16x8, w[c][2]);
#pragma unroll
      for (int t = 0; t < RT; t += 2) {
        const unsigned char* xp = xb + (16 * t) * ROWB + ((((c << 2) | lg) ^ l15) << 4);
        const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp));
        const bf16x8 xm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + PLANE));
        const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + 2 * PLANE));
        if (t + 1 < RT) {
          const unsigned char* xq = xp + 16 * ROWB;
          const bf16x8 yh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq));
          const bf16x8 ym = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + PLANE));
          const bf16x8 yl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xq + 2 * PLANE));
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, a1[t], 0, 0, 0);
          a1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, yl, a1[t + 1], 0, 0, 0);
          a0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, a0[t], 0, 0, 0);
          a0[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, yh, a0[t + 1], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, a1[t], 0, 0, 0);
          a1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, yh, a1[t + 1], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, a1[t], 0, 0, 0);
          a1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, ym, a1[t + 1], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, a1[t], 0, 0, 0);
          a1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ym, a1[t + 1], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, a1[t], 0, 0, 0);
          a1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, yh, a1[t + 1], 0, 0, 0);
        } else {
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, a1[t], 0, 0, 0);
          a0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, a0[t], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, a1[t], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, a1[t], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, a1[t], 0, 0, 0);
          a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, a1[t], 0, 0, 0);
        }
      }
    }
    unsigned char* yb = smem + (cur ^ 1) * SLOT;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const v4f s = a0[t] + a1[t];
      const float4 v = make_float4(s[0] * 1e-3f + 1.f, s[1] * 1e-3f + 1.f, s[2] * 1e-3f + 1.f, s[3] * 1e-3f + 1.f);
      uint2 H, M, L;
      split4(v, H, M, L);
      const int off = sw_off(16 * t + l15, wave * 16 + (lg << 2));
      *reinterpret_cast<uint2*>(yb + off) = H;
      *reinterpret_cast<uint2*>(yb + PLANE + off) = M;
      *reinterpret_cast<uint2*>(yb + 2 * PLANE + off) = L;
    }
    lds_barrier();
    cur ^= 1;
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (tid == 0 && smem[cur * SLOT] == 0x77) out[0] = 0;
}

template <int NW, int CT, int RS>
void run(const char* name) {
  unsigned long long* d;
  (void)hipMalloc(&d, 256 * 8 * 8);
  (void)hipMemset(d, 0, 256 * 8 * 8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&op_loop<NW, CT, RS>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
  const int iters = 50;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((op_loop<NW, CT, RS>), dim3(256), dim3(64 * NW), 2 * SLOT, 0, d, iters, 1.0f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < NW; ++w) mx = h[b * 8 + w] > mx ? (double)h[b * 8 + w] : mx;
  printf("%-58s %7.0f cycles per op (slowest wave; MFMA floor 3840)\n", name, mx / iters);
  (void)hipFree(d);
}

int main() {
  run<8, 1, 1>("A  8 waves x (16 cols, 5 row blocks)  [chain2.hip today]");
  run<4, 2, 1>("B  4 waves x (32 cols, 5 row blocks)");
  run<8, 2, 2>("C  8 waves x (32 cols, 3|2 row blocks)");
  run<8, 4, 4>("D  8 waves x (64 cols, 2|1 row blocks)");
  {
    unsigned long long* d;
    (void)hipMalloc(&d, 256 * 8 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&op_loop_pairs), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(op_loop_pairs, dim3(256), dim3(512), 2 * SLOT, 0, d, 50, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto v : h) mx = v > mx ? (double)v : mx;
    printf("%-58s %7.0f cycles per op (slowest wave; MFMA floor 3840)\n", "E  layout A, MFMAs of two row blocks interleaved", mx / 50);
    (void)hipFree(d);
  }
  return 0;
}
