// Shared device helpers for the gfx950 (CDNA4) kernels: MFMA fragment traits for the two
// compute precisions, wave64 reductions, activation functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// h16_t: the 16-bit MFMA operand type of the throughput modes.  Default build: bfloat16 (the dtype BASELINE's configs name).  The SAME
// sources compiled with -DA2P_HALF give liba2p_hip_f16.so, where h16_t is IEEE half: same MFMA rate (v_mfma_f32_16x16x32_f16),
// 3 more mantissa bits -- the rounding error of every staged operand drops 8x (precision="fp16" on the Python side; measured
// parity in profiles/r03_parity.json).  Accumulation, statistics, the residual stream and the sampler stay fp32 in both.
#ifdef A2P_HALF
typedef _Float16 h16_t;
#define A2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 h16_t;
#define A2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) h16_t h16x8;
typedef __attribute__((ext_vector_type(4))) h16_t h16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// One MFMA "k-chunk": KCH k-values per instruction, each lane holding EPL contiguous
// k-values of row/col (lane & 15), k-group (lane >> 4).  C/D layout for both:
// col = lane & 15, row = (lane >> 4) * 4 + reg   (cdna_hip_programming.md §3).
template <typename T>
struct Prec;

template <>
struct Prec<float> {  // v_mfma_f32_16x16x4_f32: exact fp32, 1/16 of the bf16 rate
  using Frag = float;
  static constexpr int KCH = 4;
  static constexpr int EPL = 1;
  __device__ static __forceinline__ f32x4 mfma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Frag load(const float* p) { return *p; }
};

template <>
struct Prec<h16_t> {  // v_mfma_f32_16x16x32_bf16
  using Frag = h16x8;
  static constexpr int KCH = 32;
  static constexpr int EPL = 8;
  __device__ static __forceinline__ f32x4 mfma(Frag a, Frag b, f32x4 c) {
    return A2P_MFMA16(a, b, c);
  }
  __device__ static __forceinline__ Frag load(const h16_t* p) { return *reinterpret_cast<const h16x8*>(p); }
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(h16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ h16_t from_f32<h16_t>(float v) { return (h16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 64 lanes without the LDS crossbar: four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, then
// row_half_mirror / row_mirror, which pair every lane with one of the OTHER quad / half once quads / halves are uniform), then
// gfx950's v_permlane16_swap / v_permlane32_swap across the rows.  ~10 VALU instructions against 6 ds_bpermute round trips; the
// summation tree differs from wave_sum's butterfly (fp32 rounding may differ in the last bit).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_valu(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror: every lane of a row now holds the row's sum
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows {0,0,2,2} and {1,1,3,3}
  const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned x = __float_as_uint(w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // halves {lo, lo} and {hi, hi}
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// activations (torch definitions)
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// bf16-mode GELU: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), v_rcp/v_exp instead of ocml erff
// (every multiply-add is an explicit fmaf: the result must not depend on which kernel instantiation inlines it)
__device__ __forceinline__ float act_gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = fmaf(-poly, __builtin_amdgcn_exp2f((-z * z) * 1.4426950408889634f), 1.0f);  // erf(|x|/sqrt2)
  const float h = 0.5f * x;
  return fmaf(h, copysignf(e, x), h);
}
__device__ __forceinline__ float act_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float act_mish(float x) { return x * tanhf(act_softplus(x)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_lrelu02(float x) { return x > 0.0f ? x : 0.2f * x; }

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_MISH = 2, ACT_SILU = 3, ACT_LRELU = 4, ACT_RELU = 5 };

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_GELU: return act_gelu(x);
    case ACT_MISH: return act_mish(x);
    case ACT_SILU: return act_silu(x);
    case ACT_LRELU: return act_lrelu02(x);
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    default: return x;
  }
}
