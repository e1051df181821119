// Which bf16 MFMA shape runs faster on THIS chip in the marching kernels' inner loop?  (MI355X_MICROARCH.md, DVFS give-back
// item 7: on random data the 16x16x32 form held a higher clock than 32x32x16 at equal cycles per FLOP.)
// Both kernels: one wave per SIMD (launch bound 1), operands re-read from LDS with ds_read_b128 in the marching kernels'
// ratio (0.75 KB of fragments per 32x32x16-equivalent MFMA), same FLOPs per iteration, accumulators kept and stored.
//   hipcc -O3 --offload-arch=gfx950 tests/diag/mfma_shape_bench.hip -o /tmp/mfma_shape_bench && /tmp/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kLds = 96 * 1024;

// group = 3 weight + 6 activation fragments, 12 MFMAs 32x32x16 on 4 accumulators (conv_march.h)
__global__ __launch_bounds__(256, 1) void k32(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < kLds / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * 977 + i) % (1 << 20)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  const char* base = smem + lane * 16 + wave * 4096;
  for (int it = 0; it < iters; ++it) {
    const char* p = base + (it & 7) * 9216;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      uint4 b[3], x[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) b[k] = *reinterpret_cast<const uint4*>(p + (g * 9 + k) * 1024 % 65536);
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = *reinterpret_cast<const uint4*>(p + (g * 9 + 3 + k) * 1024 % 65536);
#pragma unroll
      for (int hy = 0; hy < 6; ++hy)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int row = hy - kh;
          if (row >= 0 && row < 4)
            acc[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[kh]), __builtin_bit_cast(bf16x8, x[hy]), acc[row], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same tile in 16x16x32: per TWO groups (K = 32 channels per MFMA): 6 weight + 12 activation fragments, 48 MFMAs on 16
// accumulators of 4 registers (4 rows x 2 output-channel halves x 2 voxel halves)
__global__ __launch_bounds__(256, 1) void k16(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < kLds / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * 977 + i) % (1 << 20)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4][2][2];
  for (int r = 0; r < 4; ++r) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) acc[r][a][b][i] = 0.f;
  const char* base = smem + lane * 16 + wave * 4096;
  for (int it = 0; it < iters; ++it) {
    const char* p = base + (it & 7) * 9216;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      uint4 b[3][2], x[6][2];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) b[k][c] = *reinterpret_cast<const uint4*>(p + (g * 18 + k * 2 + c) * 1024 % 65536);
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int v = 0; v < 2; ++v) x[k][v] = *reinterpret_cast<const uint4*>(p + (g * 18 + 6 + k * 2 + v) * 1024 % 65536);
#pragma unroll
      for (int hy = 0; hy < 6; ++hy)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int row = hy - kh;
          if (row >= 0 && row < 4)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int v = 0; v < 2; ++v)
                acc[row][c][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[kh][c]), __builtin_bit_cast(bf16x8, x[hy][v]), acc[row][c][v], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) s += acc[r][a][b][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int n = 1 << 20;
  std::vector<uint16_t> h(n * 8);
  srand(1);
  for (auto& v : h) {                                   // bf16 N(0,1)-like: random sign, exponent around 1, random mantissa
    float f = ((rand() % 2001) - 1000) / 500.0f;
    uint32_t u; memcpy(&u, &f, 4);
    v = (uint16_t)(u >> 16);
  }
  uint4* src; float* out;
  hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 256 * 4);
  hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  for (int data = 0; data < 2; ++data) {
    if (data == 0) hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice); else hipMemset(src, 0, n * 16);
    const int iters = 400;                              // 400 x 72 MFMA-equivalents per wave: ~0.4 ms
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 3; ++round)
      for (int which = 0; which < 2; ++which) {
        for (int w = 0; w < 200; ++w) { if (which == 0) k32<<<256, 256, kLds>>>(src, out, iters); else k16<<<256, 256, kLds>>>(src, out, iters); }   // settle the clock
        hipEventRecord(e0);
        for (int w = 0; w < 50; ++w) { if (which == 0) k32<<<256, 256, kLds>>>(src, out, iters); else k16<<<256, 256, kLds>>>(src, out, iters); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 50.0 * 256 * 4 * iters * 72.0 * 32768.0;
        printf("%s data, round %d: %s  %8.1f us per launch  %7.1f TFLOP/s\n", data == 0 ? "random" : "zero  ", round,
               which == 0 ? "32x32x16" : "16x16x32", ms / 50 * 1e3, fl / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}
