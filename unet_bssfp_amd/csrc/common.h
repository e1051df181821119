// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mi355_unet.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// storage type of bf16 activations / packed weights
struct bf16_t { uint16_t v; };
// OCP e4m3 operand of the fp8 convolution path (per-tensor scaled: value * 224 / amax)
struct fp8_t { uint8_t v; };
__device__ __forceinline__ float fp8_clamp(float f) { return fminf(fmaxf(f, -448.f), 448.f); }
// two floats -> two e4m3 bytes in the low (hi = false) or high half of `old`
__device__ __forceinline__ uint32_t cvt_pk_fp8(float a, float b, uint32_t old, bool hi) {
  return hi ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(a), fp8_clamp(b), (int)old, true)
            : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(a), fp8_clamp(b), (int)old, false);
}
__device__ __forceinline__ float fp8_scale_of(const float* amax) {
  const float m = amax ? amax[0] : 0.f;
  return m > 0.f ? 224.f / m : 1.f;
}

// ---- error plumbing (thread-local message, no exceptions across the ABI) ----
void mi355_set_error(const char* fmt, ...);
int mi355_check_launch(const char* what);

#define MI355_REQUIRE(cond, ...)                      \
  do {                                                \
    if (!(cond)) {                                    \
      mi355_set_error(__VA_ARGS__);                   \
      return MI355_ERR_ARG;                           \
    }                                                 \
  } while (0)

// ---- numeric helpers ----
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, keeps NaN a NaN)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = MI355_DT_F32;
  static constexpr int kPer16B = 4;
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = MI355_DT_BF16;
  static constexpr int kPer16B = 8;
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { p->v = f32_to_bf16_bits(v); }
};
template <> struct Elem<fp8_t> {           // stores only (weight packing): the caller has applied the tensor scale
  static constexpr int kDtype = MI355_DT_FP8;
  static constexpr int kPer16B = 16;
  static __device__ __forceinline__ void store(fp8_t* p, float v) { p->v = (uint8_t)(cvt_pk_fp8(v, 0.f, 0u, false) & 0xffu); }
};

// Cache-policy hints (round 4; interleaved step A/B over diagnostic builds, profiles/r04b_ab_cache_policy.txt): data that is read
// for the last time in a pass, or written for a reader far away, takes the non-temporal policy so that it does not displace
// what the NEXT kernels re-read from L2 / the 256 MB memory-side cache.  -DMI355_DIAG_NO_NT builds the round-3 policies.
#ifndef MI355_DIAG_NO_NT
#define NORM_NT_FWD 1       // normact_fwd / normact_pool_fwd: z is dead until the backward pass                      -0.02 ms per step
#define NORM_NT_APPLY 1     // normact_bwd_apply: last reader of z and da                                            -0.08
#define WGRAD_NT_SLABS 1    // weight-gradient slabs: ~800 MB per step written for the deferred reduction            -0.065
#define ADAM_NT 1           // AdamW: gradients and moments are touched once per step                                -0.04
#define WGRAD_NT_DENSE 1    // dense slab reduction: the slabs are read once (the generic form already did, round 2)  -0.04
#endif
// (measured without effect or worse, not kept: the LDS-DMA copies of the weight-gradient kernels' operands -- their last readers
//  in the step -- with `nt` 9.921 against 9.918 ms; the f32 source of the input pack 9.913 against 9.918; the split-K slab loads of
//  conv_ksplit_reduce_kernel 10.063 against 10.057; non-temporal STORES of activations whose next reader is far away -- the skip tensor
//  written by normact_pool_fwd 10.025 against 10.013, the activation in front of the fused final convolution 9.869 against 9.886 --
//  and z streamed in normact_bwd_reduce so that da stays cached for the apply pass 9.889 against 9.886: all within noise.)
// 16-byte non-temporal load (global_load_dwordx4 ... nt): bytes that are read for the last time in this pass
__device__ __forceinline__ uint4 ld_nt_b128(const void* p) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// 16 bytes of T unpacked to floats (4 for f32, 8 for bf16)
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float f[4];
  __device__ __forceinline__ void load(const void* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  __device__ __forceinline__ void from_bits(const uint4 v) {      // 16 bytes loaded earlier (kept packed while in flight)
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ __forceinline__ void load_nt(const void* p) { from_bits(ld_nt_b128(p)); }   // last use of the bytes in this pass
  __device__ __forceinline__ void store(void* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }

};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  float f[8];
  __device__ __forceinline__ void load(const void* p) { from_bits(*reinterpret_cast<const uint4*>(p)); }
  __device__ __forceinline__ void load_nt(const void* p) { from_bits(ld_nt_b128(p)); }
  __device__ __forceinline__ void from_bits(const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(void* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)f32_to_bf16_bits(f[2 * i]) | ((uint32_t)f32_to_bf16_bits(f[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// ---- MFMA operand fragments: 16 channels of one row (voxel or output channel) ----
// f32 : two groups of 8 channels; lane half h owns channels 8g+4h .. 8g+4h+3 (one float4 each),
//       consumed by 4 x v_mfma_f32_32x32x2_f32 per group (k = {8g+s, 8g+4+s}, s = 0..3).
// bf16: lane half h owns channels 8h .. 8h+7 (one 16-B vector), one v_mfma_f32_32x32x16_bf16.
template <typename T> struct Frag;
template <> struct Frag<float> {
  float4 g0, g1;
  // p points at the 16-channel row; hoff = h*16 bytes already applied by the caller
  __device__ __forceinline__ void load(const char* p) {
    g0 = *reinterpret_cast<const float4*>(p);
    g1 = *reinterpret_cast<const float4*>(p + 32);
  }
  __device__ __forceinline__ void zero() { g0 = make_float4(0, 0, 0, 0); g1 = g0; }
};
template <> struct Frag<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const char* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
};

__device__ __forceinline__ void mma16(const Frag<float>& a, const Frag<float>& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g0.x, b.g0.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g0.y, b.g0.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g0.z, b.g0.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g0.w, b.g0.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g1.x, b.g1.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g1.y, b.g1.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g1.z, b.g1.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.g1.w, b.g1.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.v),
                                                __builtin_bit_cast(bf16x8, b.v), acc, 0, 0, 0);
}

// 16-byte streaming (non-temporal) load: data used once per workgroup should not evict re-used lines
__device__ __forceinline__ uint4 ld_stream16(const char* p) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// accumulator register i of lane half h holds D[row][col = lane & 31]
// ---- LDS-DMA (buffer_load_dwordx4 ... lds) issued as inline assembly.
// Through the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc knows the instruction writes LDS and, with one
// dynamic LDS array, cannot tell which part: it puts `s_waitcnt vmcnt(0)` in front of the NEXT ds_read, whatever
// buffer that reads -- the copy of plane p + 1 then never overlaps the MFMAs of plane p (measured: the marching kernels
// ran at the DMA's latency per plane).  As assembly the compiler does not track it; the kernel waits itself
// (dma_wait_all) before the barrier that publishes the buffer.  The compiler's own vmcnt arithmetic for its other
// memory instructions stays safe: untracked loads in flight only make a vmcnt(n) wait longer (loads return in order).
typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void* base, long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  dma_rsrc_t r = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  return r;
}
// 64 lanes x 16 bytes -> LDS bytes [lds_addr + 16 * lane, + 16); source = base + soff + voff (per lane); a source offset
// outside [0, bytes) yields zeros.
// M0: the statement overwrites m0 and cannot declare it (hipcc: "clobber list contains reserved registers: m0 ... may lead
// to undefined behaviour").  It is safe only because the kernels that use it contain NO construct for which the compiler
// itself sets or reads m0: no dynamically indexed register array (movrel-style v_movrel / s_set_gpr_idx), no
// __builtin_amdgcn_raw_ptr_buffer_load_lds, no ds_*_addtid / LDS-direct / GWS / s_sendmsg use, no readlane with a
// variable index lowered through m0.  Keep it that way in conv_march.h, conv_marchg.h and wgrad_march (wgrad.hip).
__device__ __forceinline__ void dma_lds_b128(const dma_rsrc_t rs, const void* lds_dst, const int voff, const int soff) {
  const unsigned m = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds_dst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(m), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
