// Per-voxel DTI scalar maps (reference: src/eval.py:73-135 `do_calc_scalar_maps`, with the min-max
// de-normalisation of src/eval.py:39-47 optionally fused in front).  One lane = one voxel, arithmetic in
// f64 like the reference (dti_core.h).  HBM traffic is 6 loads + 9 stores per voxel; everything else
// stays in registers.
#include "common.h"
#include "dti_core.h"

namespace {

struct DtiArgs {
  const void* t; long long nvox, cs, vs;
  double scale, offset;
  void *fa, *md, *ad, *rd, *az, *inc, *rgb;
};

template <typename T>
__global__ __launch_bounds__(256) void dti_scalar_maps_kernel(DtiArgs a) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= a.nvox) return;
  const T* t = reinterpret_cast<const T*>(a.t) + v * a.vs;
  double d[6], o[9];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = (double)t[i * a.cs] * a.scale + a.offset;
  dti_voxel_maps(d, sizeof(T) == 4, o);
  reinterpret_cast<T*>(a.fa)[v] = (T)o[0];
  reinterpret_cast<T*>(a.md)[v] = (T)o[1];
  reinterpret_cast<T*>(a.ad)[v] = (T)o[2];
  reinterpret_cast<T*>(a.rd)[v] = (T)o[3];
  reinterpret_cast<T*>(a.az)[v] = (T)o[4];
  reinterpret_cast<T*>(a.inc)[v] = (T)o[5];
  T* rgb = reinterpret_cast<T*>(a.rgb) + v * 3;
  rgb[0] = (T)o[6]; rgb[1] = (T)o[7]; rgb[2] = (T)o[8];
}

}  // namespace

extern "C" int mi355_dti_scalar_maps(const void* tensor, int32_t dtype, int64_t nvox, int64_t comp_stride,
                                     int64_t vox_stride, double scale, double offset, void* fa, void* md, void* ad,
                                     void* rd, void* azimuth, void* inclination, void* rgb, void* stream) {
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_F64, "dti_scalar_maps: dtype must be f32 or f64");
  MI355_REQUIRE(nvox >= 0 && nvox < (1ll << 40), "dti_scalar_maps: bad voxel count %lld", (long long)nvox);
  if (nvox == 0) return MI355_OK;
  MI355_REQUIRE(tensor && fa && md && ad && rd && azimuth && inclination && rgb, "dti_scalar_maps: null pointer");
  MI355_REQUIRE(comp_stride > 0 && vox_stride > 0, "dti_scalar_maps: strides must be positive");
  DtiArgs a{tensor, nvox, comp_stride, vox_stride, scale, offset, fa, md, ad, rd, azimuth, inclination, rgb};
  const unsigned blocks = (unsigned)((nvox + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MI355_DT_F32) dti_scalar_maps_kernel<float><<<blocks, 256, 0, s>>>(a);
  else dti_scalar_maps_kernel<double><<<blocks, 256, 0, s>>>(a);
  return mi355_check_launch("dti_scalar_maps");
}
