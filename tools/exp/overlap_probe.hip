// Can ONE wave keep the matrix pipe and the transcendental VALU busy at the same time on gfx950, and do two waves of a SIMD
// overlap them when they are in different phases?  The question behind "next 1" of docs/HISTORY.md section 14: the chain kernel's op
// is  MFMA phase (60 x v_mfma_f32_16x16x32_f16 per wave)  then  epilogue (20 elements per lane x (v_exp_f32 + v_rcp_f32) + ~8
// plain VALU), all eight waves of a workgroup in the same phase.  Variants, 8 waves per workgroup, one workgroup per CU:
//   A  phases in sequence (what chain2.hip does)                      B  MFMA only          C  epilogue only
//   D  per row block: the 12 MFMAs of block t+1 issued in program order in front of the epilogue of block t (compiler's schedule)
//   E  as D with sched_group_barrier: 1 MFMA, then 4 VALU/transcendental, repeated
//   F  waves 0-3 run A's order, waves 4-7 start with an epilogue (half an op out of phase, no barrier)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/overlap_probe.hip -o tools/exp/bin/overlap_probe && tools/exp/bin/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int RT = 5, KC = 4, OPS = 64;

__device__ __forceinline__ float ssilu(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e) * 1.6666666f;
}

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(float* __restrict__ out, const float* __restrict__ in) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a[KC], b[KC];
  for (int c = 0; c < KC; ++c)
    for (int i = 0; i < 8; ++i) { a[c][i] = (_Float16)in[(lane + c + i) & 255]; b[c][i] = (_Float16)in[(lane * 3 + c + i) & 255]; }
  v4f acc[RT];
  float4 v[RT];
  for (int t = 0; t < RT; ++t) { acc[t] = (v4f){0.f, 0.f, 0.f, 0.f}; v[t] = make_float4(in[lane], in[lane + 1], in[lane + 2], in[lane + 3]); }
  auto mma_block = [&](int t) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[c], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[(c + 1) & 3], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(c + 1) & 3], b[c], acc[t], 0, 0, 0);
    }
  };
  auto mma_all = [&]() {       // c outer, t inner: the order of chain2.hip (no back-to-back dependent MFMAs)
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[c], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[(c + 1) & 3], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(c + 1) & 3], b[c], acc[t], 0, 0, 0);
      }
  };
  auto epi_block = [&](int t) {
    float4 s = make_float4(acc[t][0] * 1e-3f + v[t].x, acc[t][1] * 1e-3f + v[t].y, acc[t][2] * 1e-3f + v[t].z, acc[t][3] * 1e-3f + v[t].w);
    v[t] = make_float4(ssilu(s.x), ssilu(s.y), ssilu(s.z), ssilu(s.w));
    // the accumulators start the next op from the activation's output: nothing of the MFMA phase is loop-invariant
    // (with a reset to zero the compiler hoists all 60 MFMAs out of the op loop: the first version of this probe did that)
    acc[t] = (v4f){v[t].x * 1e-3f, v[t].y * 1e-3f, v[t].z * 1e-3f, v[t].w * 1e-3f};
  };
  for (int op = 0; op < OPS; ++op) {
    if (VARIANT == 0) {
      mma_all();
#pragma unroll
      for (int t = 0; t < RT; ++t) epi_block(t);
    } else if (VARIANT == 1) {
      mma_all();
    } else if (VARIANT == 2) {
#pragma unroll
      for (int t = 0; t < RT; ++t) epi_block(t);
    } else if (VARIANT == 3 || VARIANT == 4) {
      mma_block(0);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (t + 1 < RT) mma_block(t + 1);
        epi_block(t);
        if (VARIANT == 4) {
#pragma unroll
          for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);   // 3 VALU / transcendental
          }
        }
      }
    } else {       // F: half of the waves half an op out of phase
      if (wave < 4) {
        mma_all();
#pragma unroll
        for (int t = 0; t < RT; ++t) epi_block(t);
      } else {
#pragma unroll
        for (int t = 0; t < RT; ++t) epi_block(t);
        mma_all();
      }
    }
    if (VARIANT == 0 || VARIANT == 3 || VARIANT == 4) __builtin_amdgcn_s_barrier();
  }
  float r = 0.f;
  for (int t = 0; t < RT; ++t) r += v[t].x + v[t].y + v[t].z + v[t].w + acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int V>
static void run(const char* name, float* out, const float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(256), dim3(512), 0, 0, out, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(512), 0, 0, out, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20, cyc = us * 2400.0 / OPS;
  printf("%-64s %8.1f us per launch  %7.0f cycles per op\n", name, us, cyc);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) - 0.04f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  run<0>("A  MFMA phase, then epilogue (all waves in phase, barrier)", out, in);
  run<1>("B  MFMA only (60 per wave and op)", out, in);
  run<2>("C  epilogue only (20 x (exp + rcp + 6 VALU) per lane and op)", out, in);
  run<3>("D  block t+1's MFMAs in front of block t's epilogue", out, in);
  run<4>("E  D + sched_group_barrier (1 MFMA : 3 VALU)", out, in);
  run<5>("F  A, half of the waves half an op out of phase (no barrier)", out, in);
  return 0;
}
