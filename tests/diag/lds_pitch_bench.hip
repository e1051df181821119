// LDS bank-conflict probe for ds_read_b128 fragment reads (gfx950): lane (r, h) of a wave reads 16 bytes of "voxel" r of a row,
// half h -- the access of the MFMA activation fragments.  Layouts: pitch P bytes per voxel, optional XOR swizzle of the 16-byte
// piece.  Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; one kernel per layout.
// build: hipcc -O3 --offload-arch=gfx950 tests/diag/lds_pitch_bench.hip -o tests/diag/lds_pitch_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[48 * 1024];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  int off;
  if (MODE == 0) off = r * 48 + h * 16;                                   // 48-byte pitch, 32 consecutive voxels
  else if (MODE == 1) off = ((r >> 4) * 18 + (r & 15)) * 48 + h * 16;     // 48-byte pitch, two rows of 16 voxels (row pitch 18 voxels)
  else if (MODE == 2) off = r * 32 + ((h ^ ((r >> 3) & 1)) * 16);         // 32-byte pitch, half swizzled by voxel bit 3
  else if (MODE == 3) off = r * 32 + h * 16;                              // 32-byte pitch, plain
  else if (MODE == 4) off = (r * 4 + ((2 * 0 + h) ^ ((r >> 2) & 3))) * 16; // 64-byte pitch, conv_marchg swizzle (chunk 0)
  else if (MODE == 5) off = r * 80 + h * 16;                              // 80-byte pitch
  else if (MODE == 6) off = (h * 32 + r) * 16;                            // fragment-contiguous (lane * 16)
  else off = ((r >> 4) * 18 + (r & 15)) * 32 + ((h ^ ((((r >> 4) * 18 + (r & 15)) >> 3) & 1)) * 16);   // 32-byte pitch swizzled, two rows
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint4 v = *reinterpret_cast<const uint4*>(smem + off + (it & 7) * 2304);
    acc += __uint_as_float(v.x) + __uint_as_float(v.w);
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  const int iters = 4096;
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
