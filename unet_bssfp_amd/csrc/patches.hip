// Sliding-window inference data movement (reference: src/model.py:291-333 with TorchIO's GridSampler /
// GridAggregator, src/data_module.py:168-183): gather a batch of patches out of a [C][D][H][W] volume
// and aggregate predicted patches back into one.  f32, NCDHW on both sides (what Generator.forward takes
// and returns).  The aggregation is written from the OUTPUT voxel's point of view -- one lane owns one
// voxel and walks the patches of the launch in sampler order -- so overlapping patches need no atomics
// and the result does not depend on scheduling: "crop" = the last patch covering the voxel wins (the
// reference's assignment order), "average" = sum and count accumulate in sampler order.
#include "common.h"

namespace {

struct PatchLocs { int n; int v[MI355_MAX_PATCHES][9]; };   // origin[3], kept region ini[3], fin[3] (volume coords)

__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ vol, float* __restrict__ out,
                                                           int c, int d, int h, int w, int pd, int ph, int pw, PatchLocs L) {
  const long long pv = (long long)pd * ph * pw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pv) return;
  const int b = blockIdx.z, ch = blockIdx.y;
  const int z = (int)(i / ((long long)ph * pw)), y = (int)(i / pw % ph), x = (int)(i % pw);
  const long long src = (((long long)ch * d + (L.v[b][0] + z)) * h + (L.v[b][1] + y)) * w + (L.v[b][2] + x);
  out[((long long)b * c + ch) * pv + i] = vol[src];
}

template <int MODE>   // 0 crop (overwrite), 1 average (accumulate sum + count)
__global__ __launch_bounds__(256) void patch_aggregate_kernel(const float* __restrict__ patches, float* __restrict__ vol,
                                                              float* __restrict__ count, int c, int d, int h, int w,
                                                              int pd, int ph, int pw, PatchLocs L) {
  const long long sv = (long long)d * h * w;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= sv) return;
  const int ch = blockIdx.y;
  const int z = (int)(i / ((long long)h * w)), y = (int)(i / w % h), x = (int)(i % w);
  const long long pv = (long long)pd * ph * pw;
  // average: continue the running sum exactly where earlier launches left it (sequential sampler order)
  float acc = MODE == 1 ? vol[(long long)ch * sv + i] : 0.f, cnt = 0.f;
  int last = -1;
  for (int b = 0; b < L.n; ++b) {                       // uniform trip count, locations live in SGPRs
    const int* l = L.v[b];
    if (z >= l[3] && z < l[6] && y >= l[4] && y < l[7] && x >= l[5] && x < l[8]) {
      if (MODE == 0) last = b;
      else {
        acc += patches[((long long)b * c + ch) * pv + ((long long)(z - l[0]) * ph + (y - l[1])) * pw + (x - l[2])];
        cnt += 1.f;
      }
    }
  }
  if (MODE == 0) {
    if (last >= 0) {
      const int* l = L.v[last];
      vol[(long long)ch * sv + i] = patches[((long long)last * c + ch) * pv + ((long long)(z - l[0]) * ph + (y - l[1])) * pw + (x - l[2])];
    }
  } else if (cnt > 0.f) {
    vol[(long long)ch * sv + i] = acc;
    if (ch == 0) count[i] += cnt;
  }
}

__global__ __launch_bounds__(256) void patch_average_finalize_kernel(float* __restrict__ vol, const float* __restrict__ count,
                                                                     long long sv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= sv) return;
  vol[(long long)blockIdx.y * sv + i] /= count[i];     // 0/0 -> NaN where no patch landed, like torch's true-divide
}

int fill_locs(PatchLocs& L, const int32_t* locs, int n, int stride, int d, int h, int w, int pd, int ph, int pw, bool kept) {
  L.n = n;
  const int dims[3] = {d, h, w}, ps[3] = {pd, ph, pw};
  for (int b = 0; b < n; ++b) {
    const int32_t* l = locs + (long long)b * stride;
    for (int k = 0; k < 3; ++k) {
      MI355_REQUIRE(l[k] >= 0 && l[k] + ps[k] <= dims[k], "patches: patch %d leaves the volume (axis %d: %d + %d > %d)", b, k, l[k], ps[k], dims[k]);
      L.v[b][k] = l[k];
      const int ini = kept ? l[3 + k] : l[k], fin = kept ? l[6 + k] : l[k] + ps[k];
      MI355_REQUIRE(ini >= l[k] && fin <= l[k] + ps[k] && ini <= fin, "patches: kept region of patch %d is outside the patch", b);
      L.v[b][3 + k] = ini; L.v[b][6 + k] = fin;
    }
  }
  return MI355_OK;
}

}  // namespace

extern "C" int mi355_patch_gather(const float* vol, int32_t c, int32_t d, int32_t h, int32_t w, const int32_t* origins,
                                  int32_t npatches, int32_t pd, int32_t ph, int32_t pw, float* out, void* stream) {
  MI355_REQUIRE(c > 0 && d > 0 && h > 0 && w > 0 && pd > 0 && ph > 0 && pw > 0 && c <= 65535, "patch_gather: bad shape");
  MI355_REQUIRE(npatches >= 0, "patch_gather: bad patch count");
  if (npatches == 0) return MI355_OK;
  MI355_REQUIRE(vol && origins && out, "patch_gather: null pointer");
  const long long pv = (long long)pd * ph * pw;
  for (int b0 = 0; b0 < npatches; b0 += MI355_MAX_PATCHES) {
    const int n = npatches - b0 < MI355_MAX_PATCHES ? npatches - b0 : MI355_MAX_PATCHES;
    PatchLocs L;
    const int rc = fill_locs(L, origins + (long long)b0 * 3, n, 3, d, h, w, pd, ph, pw, false);
    if (rc != MI355_OK) return rc;
    patch_gather_kernel<<<dim3((unsigned)((pv + 255) / 256), c, n), 256, 0, (hipStream_t)stream>>>(
        vol, out + (long long)b0 * c * pv, c, d, h, w, pd, ph, pw, L);
  }
  return mi355_check_launch("patch_gather");
}

extern "C" int mi355_patch_aggregate(const float* patches, const int32_t* locs9, int32_t npatches, int32_t pd, int32_t ph,
                                     int32_t pw, int32_t mode, float* vol, float* count, int32_t c, int32_t d, int32_t h,
                                     int32_t w, void* stream) {
  MI355_REQUIRE(c > 0 && d > 0 && h > 0 && w > 0 && pd > 0 && ph > 0 && pw > 0 && c <= 65535, "patch_aggregate: bad shape");
  MI355_REQUIRE(mode == MI355_AGG_CROP || mode == MI355_AGG_AVERAGE, "patch_aggregate: unknown mode %d", mode);
  MI355_REQUIRE(npatches >= 0, "patch_aggregate: bad patch count");
  if (npatches == 0) return MI355_OK;
  MI355_REQUIRE(patches && locs9 && vol && (mode == MI355_AGG_CROP || count), "patch_aggregate: null pointer");
  const long long sv = (long long)d * h * w, pv = (long long)pd * ph * pw;
  for (int b0 = 0; b0 < npatches; b0 += MI355_MAX_PATCHES) {   // chunks run in stream order: later chunks overwrite earlier
    const int n = npatches - b0 < MI355_MAX_PATCHES ? npatches - b0 : MI355_MAX_PATCHES;
    PatchLocs L;
    const int rc = fill_locs(L, locs9 + (long long)b0 * 9, n, 9, d, h, w, pd, ph, pw, true);
    if (rc != MI355_OK) return rc;
    const dim3 grid((unsigned)((sv + 255) / 256), c);
    const float* p = patches + (long long)b0 * c * pv;
    if (mode == MI355_AGG_CROP) patch_aggregate_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(p, vol, count, c, d, h, w, pd, ph, pw, L);
    else patch_aggregate_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(p, vol, count, c, d, h, w, pd, ph, pw, L);
  }
  return mi355_check_launch("patch_aggregate");
}

extern "C" int mi355_patch_average_finalize(float* vol, const float* count, int32_t c, int64_t voxels, void* stream) {
  MI355_REQUIRE(vol && count && c > 0 && c <= 65535 && voxels > 0, "patch_average_finalize: bad argument");
  patch_average_finalize_kernel<<<dim3((unsigned)((voxels + 255) / 256), c), 256, 0, (hipStream_t)stream>>>(vol, count, voxels);
  return mi355_check_launch("patch_average_finalize");
}
