// LDS bank-conflict probe, read side and write side apart (conv_lowg_kernel reads 42 % on
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE while conv_marchg_kernel, which fills LDS by LDS-DMA, reads 0 on the same kind of
// ds_read_b128 fragment reads -- which side produces it?).  One workgroup per CU (100 KB of LDS), 4 waves, as conv_lowg_kernel.
//   R48   ds_read_b128, lane (r, h) -> r * 48 + h * 16            conv_lowg's fragment reads (48-byte pitch)
//   W48   ds_write_b128, piece p = tid -> (p >> 1) * 48 + (p & 1) * 16   conv_lowg's halo / weight-slot writes (round 3)
//   W48R  ds_write_b128, lane j of a 32-lane half -> voxel (j & 15), half (j >> 4)   the same bytes, lanes permuted so that 16
//         consecutive lanes write 16 different 16-byte bank groups (the read pattern's property)
//   R48x2 / R48x4  conv_lowg's HALO fragment reads: 32 voxels as 2 rows of 16 (tile 4x8x16) / 4 rows of 8 (tile 8x8x8)
//   RL16  ds_read_b128, lane * 16 (the trivially conflict-free reference)      WL16  ds_write_b128, lane * 16
// Prints cycles (s_memtime) per instruction of wave 0 of block 0; run it a second time under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace
// for the counters per kernel.  build: hipcc -O3 --offload-arch=gfx950 tests/diag/lds_rw_bench.hip -o tests/diag/lds_rw_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int kLds = 100 * 1024;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < kLds / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  int off;
  if (MODE == 0) off = r * 48 + h * 16 + wave * 3072;
  else if (MODE == 1) off = (tid >> 1) * 48 + (tid & 1) * 16;
  else if (MODE == 2) { const int j = tid & 31, g = tid >> 5; off = (g * 16 + (j & 15)) * 48 + (j >> 4) * 16; }
  else if (MODE == 5) off = ((r >> 4) * 18 + (r & 15)) * 48 + h * 16 + wave * 3072;   // two rows of 16 voxels, halo row pitch 18 (TW = 16)
  else if (MODE == 6) off = ((r >> 3) * 10 + (r & 7)) * 48 + h * 16 + wave * 3072;     // four rows of 8 voxels, halo row pitch 10 (TW = 8)
  else off = tid * 16;
  constexpr bool READ = MODE == 0 || MODE == 3 || MODE == 5 || MODE == 6;
  u32x4 v = {(unsigned)tid, (unsigned)tid + 1, (unsigned)tid + 2, (unsigned)tid + 3};
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 4) {
    // (inline asm: from C++ hipcc narrows a 16-byte LDS read of which two words are used to ds_read2_b32 -- what round 3's
    //  lds_pitch_bench.hip measured without noticing.)  Four instructions in flight per wave, 16 per CU: the LDS pipe is the limit.
    const unsigned addr = (unsigned)(size_t)(smem + off + (it & 4) * 6144);
    if (READ) {
      u32x4 x0, x1, x2, x3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:6144\n\tds_read_b128 %2, %4 offset:12288\n\t"
                   "ds_read_b128 %3, %4 offset:18432\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(addr) : "memory");
      acc += __uint_as_float(x0.x ^ x1.y ^ x2.z ^ x3.w);
    } else {
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:6144\n\tds_write_b128 %0, %1 offset:12288\n\t"
                   "ds_write_b128 %0, %1 offset:18432\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
      v.x += 1;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  acc += reinterpret_cast<float*>(smem)[tid];
  out[blockIdx.x * 256 + tid] = acc;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc, int iters) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), kLds, 0, out, cyc, iters);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), kLds, 0, out, cyc, iters);
  long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-5s %8.2f cycles per instruction (wave 0, 4 waves per CU issuing)\n", name, (double)c / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 8192;
  run<0>("R48", out, cyc, iters);
  run<1>("W48", out, cyc, iters);
  run<2>("W48R", out, cyc, iters);
  run<3>("RL16", out, cyc, iters);
  run<4>("WL16", out, cyc, iters);
  run<5>("R48x2", out, cyc, iters);
  run<6>("R48x4", out, cyc, iters);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
