// Host build of the kernel's per-voxel arithmetic (csrc/dti_core.h) for the CPU test-suite:
// reads n voxels x 6 doubles from stdin-named file, writes n x 9 doubles.  Test harness only.
#include <stdio.h>
#include <stdlib.h>
#include "../unet_bssfp_amd/csrc/dti_core.h"
int main(int argc, char** argv) {
  if (argc != 5) return 2;
  const long n = atol(argv[3]); const int f32_angles = atoi(argv[4]);
  double* in = (double*)malloc(sizeof(double) * 6 * n); double* out = (double*)malloc(sizeof(double) * 9 * n);
  FILE* fi = fopen(argv[1], "rb"); if (!fi || fread(in, sizeof(double) * 6, n, fi) != (size_t)n) return 3; fclose(fi);
  for (long i = 0; i < n; ++i) dti_voxel_maps(in + 6 * i, f32_angles != 0, out + 9 * i);
  FILE* fo = fopen(argv[2], "wb"); if (!fo || fwrite(out, sizeof(double) * 9, n, fo) != (size_t)n) return 4; fclose(fo);
  return 0;
}
