// ds_read_b128 throughput of the chain2 fragment-read pattern vs a linear pattern (diagnosis, not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/lds_read_bench.hip -o tools/exp/bin/lds_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ROWB = 256, RT = 5, PLANE = 16 * RT * ROWB, SLOT = 3 * PLANE;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 2 * SLOT / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int tb = 0; tb < RT; tb += DEPTH) {
        uint4 x[DEPTH][3];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
          const int t = tb + u;
          if (t < RT) {
            const unsigned char* p;
            if (PAT == 0) p = smem + (16 * t + l15) * ROWB + ((((c << 2) | lg) ^ l15) << 4);      // chain2: swizzled fragment
            else if (PAT == 1) p = smem + (16 * t) * ROWB + c * 1024 + lane * 16;                 // linear 1 KB
            else if (PAT == 2) p = smem + (16 * t + l15) * ROWB + (((c << 2) | lg) << 4);         // unswizzled fragment (16-way)
            else p = smem + (16 * t + l15) * ROWB + ((((c << 2) | lg) ^ (l15 >> 1)) << 4);        // half swizzle (2-way)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) x[u][pl] = *reinterpret_cast<const uint4*>(p + pl * PLANE);
          }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
          if (tb + u < RT) for (int pl = 0; pl < 3; ++pl) acc += x[u][pl].x ^ x[u][pl].y ^ x[u][pl].z ^ x[u][pl].w;
      }
    }
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0 + (acc == 0x12345 ? 1 : 0);
}

// the same bytes as ds_read_b64 pairs: lane reads 8 B from each of two 512 B lines (linear), or two halves of its fragment
template <int PAT>
__global__ __launch_bounds__(512) void k64(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 2 * SLOT / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned char* p;
          int second;
          if (PAT == 0) { p = smem + pl * PLANE + (16 * t) * ROWB + c * 1024 + lane * 8; second = 512; }
          else { p = smem + pl * PLANE + (16 * t + l15) * ROWB + ((((c << 2) | lg) ^ l15) << 4); second = 8; }
          const uint2 a = *reinterpret_cast<const uint2*>(p);
          const uint2 b = *reinterpret_cast<const uint2*>(p + second);
          acc += a.x ^ a.y ^ b.x ^ b.y;
        }
      }
    }
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0 + (acc == 0x12345 ? 1 : 0);
}

template <int PAT>
void run64(const char* name, int nthreads) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 8);
  hipMemset(d, 0, 256 * 8 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k64<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
  const int iters = 20;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k64<PAT>), dim3(256), dim3(nthreads), 2 * SLOT, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const int nw = nthreads / 64;
  double mx = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { double v = (double)h[b * 8 + w]; if (v > mx) mx = v; }
  const double per_op = mx / iters, bytes = (double)nw * 60 * 1024;
  printf("%-28s waves %d b64 pairs: %7.0f cycles per op (slowest wave), %6.1f B/clk/CU\n", name, nw, per_op, bytes / per_op);
  hipFree(d);
}

template <int PAT, int DEPTH>
void run(const char* name, int nthreads) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 8);
  hipMemset(d, 0, 256 * 8 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PAT, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
  const int iters = 20;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<PAT, DEPTH>), dim3(256), dim3(nthreads), 2 * SLOT, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const int nw = nthreads / 64;
  double mx = 0, sum = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { double v = (double)h[b * 8 + w]; sum += v; if (v > mx) mx = v; }
  const double per_op = mx / iters, bytes = (double)nw * 60 * 1024;
  printf("%-28s waves %d depth %d: %7.0f cycles per 60-read op (slowest wave), %6.1f B/clk/CU\n", name, nw, DEPTH, per_op, bytes / per_op);
  hipFree(d);
}

int main() {
  run<0, 1>("chain2 swizzled", 512); run<0, 3>("chain2 swizzled", 512); run<0, 5>("chain2 swizzled", 512);
  run<0, 1>("chain2 swizzled", 256); run<0, 5>("chain2 swizzled", 256);
  run<1, 1>("linear", 512); run<1, 3>("linear", 512); run<1, 5>("linear", 512); run<1, 5>("linear", 256);
  run64<0>("linear b64", 512); run64<0>("linear b64", 256); run64<1>("fragment halves b64", 512);
  run<3, 5>("half swizzle (2-way)", 512);
  run<2, 5>("unswizzled (16-way)", 512);
  return 0;
}
