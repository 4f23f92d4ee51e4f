// HBM-bound kernels of the denoiser / sampler: LayerNorm + full-width rotary, casts / packing,
// the per-step time path glue, classifier-free-guidance combine + DDIM/DDPM posterior update.
#pragma once
#include "a2p_common.h"

// ---------------------------------------------------------------------------------------------
// cos/sin table of the rotary embedding: cs[pos][i] = (cos(pos*f_i), sin(pos*f_i)), angle formed in
// fp32 like the reference's cached table (rotary_embedding_torch.py:124-139).
// ---------------------------------------------------------------------------------------------
__global__ void rope_table_kernel(const float* __restrict__ freqs, float2* __restrict__ cs, int npos, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos * half) return;
  const int pos = i / half, k = i - pos * half;
  const float ang = (float)pos * freqs[k];
  cs[i] = make_float2(cosf(ang), sinf(ang));
}

// The same table for the chain kernels, which read it one row-panel at a time: [d/4][npos] entries of 16 bytes = (cos, sin) of
// the two pairs of 4 consecutive columns, positions contiguous.  A chain lane owns (row = lane & 15, 4 columns), so one load
// instruction then touches 4 x 256 contiguous bytes instead of 64 separate 16-byte segments of the [pos][d/2] layout.
__global__ void rope_table_t_kernel(const float2* __restrict__ cs, float4* __restrict__ cst, int npos, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos * (half / 2)) return;
  const int q = i / npos, pos = i - q * npos;
  const float2 a = cs[(size_t)pos * half + 2 * q], b = cs[(size_t)pos * half + 2 * q + 1];
  cst[i] = make_float4(a.x, a.y, b.x, b.y);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5, biased variance) over d = 64*NPL, optionally followed by the full-width
// interleaved-pair rotation (rotary_embedding_torch.py:46-66).  One wave per row, fp32 statistics.
//   xn = LN(x)           -> out_n (dtype T), may be NULL
//   xr = rotary(LN(x))   -> out_r (dtype T), may be NULL, position = row % rows_per_seq + pos_off
// gamma == NULL skips the normalisation (rotation only, used for the keyframe tokens).
// ---------------------------------------------------------------------------------------------
struct LnRopeP {
  const float* x;
  int64_t ldx;
  const float* gamma;
  const float* beta;
  const float2* cs;  // [pos][d/2]
  void* out_n;
  void* out_r;
  int64_t ldo;
  int rows, rows_per_seq, pos_off;
};

template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_rope_kernel(LnRopeP p) {
  constexpr int D = 64 * NPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int e0 = lane * NPL;
  float v[NPL];
  const float* xr = p.x + (int64_t)row * p.ldx + e0;
#pragma unroll
  for (int i = 0; i < NPL; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(xr + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
  if (p.gamma) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      v[i] -= mean;
      q += v[i] * v[i];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = v[i] * rstd * p.gamma[e0 + i] + p.beta[e0 + i];
  }
  if (p.out_n) {
    T* o = reinterpret_cast<T*>(p.out_n) + (int64_t)row * p.ldo + e0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) o[i] = from_f32<T>(v[i]);
  }
  if (p.out_r) {
    const int pos = row % p.rows_per_seq + p.pos_off;
    const float2* c = p.cs + (int64_t)pos * (D / 2) + e0 / 2;
    T* o = reinterpret_cast<T*>(p.out_r) + (int64_t)row * p.ldo + e0;
#pragma unroll
    for (int i = 0; i < NPL; i += 2) {
      const float2 t = c[i / 2];
      o[i] = from_f32<T>(v[i] * t.x - v[i + 1] * t.y);
      o[i + 1] = from_f32<T>(v[i + 1] * t.x + v[i] * t.y);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 [rows, cols] -> T [rows, cols_pad] (zero pad); optional per-row keep mask (keyframes:
// "pad the unknown", model/diffusion.py:319-320).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ src, int64_t lds, int src_col_stride, T* __restrict__ dst, int64_t ldd,
                                int64_t rows, int cols, int cols_pad, const uint8_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols_pad) return;
  const int64_t r = i / cols_pad;
  const int c = (int)(i - r * cols_pad);
  float v = 0.f;
  if (c < cols && (!keep || keep[r])) v = src[r * lds + (int64_t)c * src_col_stride];
  dst[r * ldd + c] = from_f32<T>(v);
}

// x [B, C, T] fp32 -> A [B*T, Cpad] T  (the permute of model/diffusion.py:345-346 fused with the cast)
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, T* __restrict__ dst, int B, int C, int Tn, int Cpad) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < Tn) ? x[((int64_t)b * C + c) * Tn + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < Tn && c < Cpad) dst[((int64_t)b * Tn + t) * Cpad + c] = from_f32<T>(tile[tx][i]);
  }
}

// ---------------------------------------------------------------------------------------------
// Split operands ("x3"): an fp32 value v as the 16-bit pair hi = T(v), lo = T(v - hi) carries 2 x (mantissa bits of T) bits, and
//     a . w  =  a_hi w_hi + a_lo w_hi + a_hi w_lo  (+ a_lo w_lo, 2^-22 relative: dropped)
// is ONE 16-bit GEMM over a 3x longer contraction: A' = [a_hi | a_lo | a_hi], W' = [w_hi | w_hi | w_lo].  The 16-bit modes run
// input_projection, final_layer and the pose conv tail this way: those operand classes carry the error of the sampling loop's
// return value (profiles/r03_error_budget*.json), exact fp32 MFMA (1/16 of the 16-bit rate) cost the body model 14 %.
// ---------------------------------------------------------------------------------------------
// fp32 [rows][cols] (row stride lds, column stride src_col_stride) -> T [rows][3 * cpad]; weights: [hi | hi | lo]
__global__ void split3_kernel(const float* __restrict__ src, int64_t lds, int src_col_stride, h16_t* __restrict__ dst, int64_t rows,
                              int cols, int cpad, int weight) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cpad) return;
  const int64_t r = i / cpad;
  const int c = (int)(i - r * cpad);
  const float v = c < cols ? src[r * lds + (int64_t)c * src_col_stride] : 0.f;
  const h16_t hi = (h16_t)v, lo = (h16_t)(v - (float)hi);
  h16_t* d = dst + r * 3 * cpad + c;
  d[0] = hi;
  d[cpad] = weight ? hi : lo;
  d[2 * cpad] = weight ? lo : hi;
}

// x [B, C, T] fp32 -> A' [B*T, 3 * Cpad] split operand rows (the permute of model/diffusion.py:345-346 fused with the split)
__global__ void pack_input_split3_kernel(const float* __restrict__ x, h16_t* __restrict__ dst, int B, int C, int Tn, int Cpad) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < Tn) ? x[((int64_t)b * C + c) * Tn + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < Tn && c < Cpad) {
      const float v = tile[tx][i];
      const h16_t hi = (h16_t)v, lo = (h16_t)(v - (float)hi);
      h16_t* d = dst + ((int64_t)b * Tn + t) * 3 * Cpad + c;
      d[0] = hi; d[Cpad] = lo; d[2 * Cpad] = hi;
    }
  }
}

// mean over the token axis: src fp32 [B, S, d] -> dst [B, d]   (model/diffusion.py:380)
// One workgroup = 64 columns x 16 row groups; a thread sums the rows rg, rg + 16, ... of its column (four independent partial
// sums: the loads of a thread do not depend on one another), the 16 partials meet in LDS in a fixed order (deterministic).
// Round 3 walked all 1998 rows with ONE thread per column: 475 us per call, 60 % of a2p_prepare_cond.
__global__ __launch_bounds__(1024) void mean_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int S, int d) {
  __shared__ float part[16][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < d) {
    const float* p = src + (int64_t)b * S * d + c;
    int i = rg;
    for (; i + 48 < S; i += 64) {
      s0 += p[(int64_t)i * d];
      s1 += p[(int64_t)(i + 16) * d];
      s2 += p[(int64_t)(i + 32) * d];
      s3 += p[(int64_t)(i + 48) * d];
    }
    for (; i < S; i += 16) s0 += p[(int64_t)i * d];
  }
  part[rg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < d) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += part[r][lane];
    dst[(int64_t)b * d + c] = s / (float)S;
  }
}

// SinusoidalPosEmb (model/utils.py:67-79): emb[b] = [sin(t f_k), cos(t f_k)], f_k host-built fp32.
__global__ void time_embed_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq, float* __restrict__ emb,
                                  int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float a = (float)t[b] * freq[k];
  emb[(int64_t)b * 2 * half + k] = sinf(a);
  emb[(int64_t)b * 2 * half + half + k] = cosf(a);
}

// Glue of the time path (model/diffusion.py:385-393): per sequence t = to_time_cond(h) + cond_hidden
// (null_cond_hidden for unconditional sequences), mish(t) for the FiLM generators
// (transformer_modules.py:111-113), and norm_cond + rotary of the two time tokens.
struct TPathP {
  const float* tct;     // [B, 3d]: time cond | token0 | token1
  const float* hidden;  // [(B+1), d] slot table (0 = null_cond_hidden)
  const int* slot;      // [nseq]
  float* tvec;          // [nseq, d]
  float* mt;            // [nseq, d]  mish(t)
  const float* gamma;   // norm_cond
  const float* beta;
  const float2* cs;
  float* tok_n;         // [B*2, d]
  float* tok_r;         // [B*2, d]
  int B, nseq, d, pos0; // pos0 = number of audio tokens (time tokens sit at pos0, pos0+1)
  // Time-MLP table (a2p_ctx::tct_table): tct holds one row per TIMESTEP VALUE 0 .. trows-1 (time_embed -> time_mlp -> to_time_cond | to_time_tokens depend on t alone),
  // trow[b] = sample b's timestep selects the row.  A timestep outside the table sets bit 2 of *err (a2p_check_finite reports it) and reads row 0 / trows-1.
  const int64_t* trow;
  int trows;
  int* err;
};
__device__ __forceinline__ int64_t tpath_row(const TPathP& p, int b) {
  if (!p.trow) return b;
  int64_t t = p.trow[b];
  if (t < 0 || t >= p.trows) {
    if (threadIdx.x == 0 && p.err) atomicOr(p.err, 4);
    t = t < 0 ? 0 : p.trows - 1;
  }
  return t;
}

template <int NPL>  // d / 64: 8 (face) or 4 (pose) contiguous features per lane in the token part
__global__ __launch_bounds__(256) void tpath_post_kernel(TPathP p) {
  const int d = p.d;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x;
  if (blk < p.nseq) {  // t vector of one sequence
    const int n = blk, b = n % p.B, sl = p.slot[n];
    const int64_t tr = tpath_row(p, b);
    for (int c = threadIdx.x; c < d; c += 256) {
      const float v = p.tct[tr * 3 * d + c] + p.hidden[(int64_t)sl * d + c];
      p.tvec[(int64_t)n * d + c] = v;
      p.mt[(int64_t)n * d + c] = act_mish(v);
    }
    return;
  }
  // time tokens of one sample: waves 0,1 -> token 0,1
  const int b = blk - p.nseq;
  if (wid >= 2) return;
  const float* src = p.tct + tpath_row(p, b) * 3 * d + d + wid * d;
  const int row = b * 2 + wid, pos = p.pos0 + wid;
  // The rotary (cos, sin) pairs are fetched first and pinned in registers: they are not consumed until after both
  // LayerNorm reductions, so the loads have long landed by then.  (With the fetch next to its first use, the first VALU
  // after the s_waitcnt -- a packed f32 op selecting the high dword of the returning dwordx2 -- was observed to read 0
  // in lanes 48-63 when another stream's kernel loaded the memory system; see docs/lab_notebook_r1_r4.md section 6, "Reproducibility".)
  float2 t[NPL / 2];
#pragma unroll
  for (int i = 0; i < NPL / 2; ++i) t[i] = p.cs[(int64_t)pos * (d / 2) + lane * (NPL / 2) + i];
#pragma unroll
  for (int i = 0; i < NPL / 2; ++i) asm volatile("" : "+v"(t[i].x), "+v"(t[i].y));
  float v[NPL], g[NPL], be[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    v[i] = src[lane * NPL + i];
    g[i] = p.gamma[lane * NPL + i];
    be[i] = p.beta[lane * NPL + i];
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) asm volatile("" : "+v"(v[i]), "+v"(g[i]), "+v"(be[i]));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) s += v[i];
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    v[i] -= mean;
    q += v[i] * v[i];
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / d + 1e-5f);
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    v[i] = v[i] * rstd * g[i] + be[i];
    p.tok_n[(int64_t)row * d + lane * NPL + i] = v[i];
  }
#pragma unroll
  for (int i = 0; i < NPL / 2; ++i) {
    const int c = lane * NPL + 2 * i;
    p.tok_r[(int64_t)row * d + c] = v[2 * i] * t[i].x - v[2 * i + 1] * t[i].y;
    p.tok_r[(int64_t)row * d + c + 1] = v[2 * i + 1] * t[i].x + v[2 * i] * t[i].y;
  }
}

__global__ void mish_kernel(const float* __restrict__ a, float* __restrict__ o, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = act_mish(a[i]);
}

// ---------------------------------------------------------------------------------------------
// Sampler arithmetic.  Tables are fp32 rows of length n_steps (a2p_table_id order); all formulas in
// fp32 in the reference's operation order (gaussian_diffusion.py:235-257, 305-316, 347-351,
// 470-476, 699-717).
// ---------------------------------------------------------------------------------------------
enum { TAB_C1 = 0, TAB_C2, TAB_VAR, TAB_LOGVAR, TAB_SRA, TAB_SRM1, TAB_ACP, TAB_ACPP, TAB_SQRT_ACP, TAB_SQRT_1M, TAB_ACPN };

__device__ __forceinline__ float ddim_update(float x0, float x, float noise, const float* tab, int ns, int t, float eta) {
  const float eps = (tab[TAB_SRA * ns + t] * x - x0) / tab[TAB_SRM1 * ns + t];
  const float ab = tab[TAB_ACP * ns + t], abp = tab[TAB_ACPP * ns + t];
  const float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
  const float mean_pred = x0 * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * eps;
  const float nz = t != 0 ? 1.f : 0.f;
  return mean_pred + nz * sigma * noise;
}

__device__ __forceinline__ float ddpm_update(float x0, float x, float noise, const float* tab, int ns, int t) {
  const float mean = tab[TAB_C1 * ns + t] * x0 + tab[TAB_C2 * ns + t] * x;
  const float nz = t != 0 ? 1.f : 0.f;
  return mean + nz * expf(0.5f * tab[TAB_LOGVAR * ns + t]) * noise;
}

// Fused tail of one sampling step: classifier-free-guidance combine (model/cfg_sampler.py:33),
// [B,T,C] -> [B,C,1,T] (gaussian_diffusion.py:312-313), optional clamp, posterior update.
struct StepP {
  const float* mo;        // model output rows: mo[(seq*mo_seq_rows + t) * mo_ld + c]
  int64_t mo_seq_rows, mo_ld;
  int B, C, Tn;
  int pass;               // A2P_PASS_*: CFG reads sequences b and B+b
  const float* scale;     // [B]
  float* out_btc;         // [B,T,C] or NULL
  int sampler;            // -1 none, 0 ddim, 1 ddpm
  const float* x;         // [B,C,T]
  const int64_t* t_idx;   // [B]
  const float* tables;
  int n_steps;
  const float* noise;     // [B,C,T] or NULL
  float eta;
  int clip;
  float* x_next;          // [B,C,T]
  float* x0;              // [B,C,T] pred_xstart
  float* mean;            // optional posterior mean [B,C,T] (p_mean_variance API)
  int* nonfinite;         // optional device flag: OR-ed with 1 when a model output element is inf / nan (a2p_check_finite)
};

__global__ __launch_bounds__(256) void step_tail_kernel(StepP p) {
  __shared__ float tile[32][33];  // [t][c]
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float sc = (p.pass == 2) ? p.scale[b] : 0.f;
  bool bad = false;
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    float g = 0.f;
    if (t < p.Tn && c < p.C) {
      const float a = p.mo[((int64_t)b * p.mo_seq_rows + t) * p.mo_ld + c];
      if (p.pass == 2) {
        const float u = p.mo[((int64_t)(p.B + b) * p.mo_seq_rows + t) * p.mo_ld + c];
        g = u + sc * (a - u);
      } else {
        g = a;
      }
      bad |= !(fabsf(g) <= 3.4028234e38f);   // inf or nan (a nan fails every comparison); inf - inf of the two passes is nan
      if (p.out_btc) p.out_btc[((int64_t)b * p.Tn + t) * p.C + c] = g;
    }
    tile[i][tx] = g;
  }
  if (bad && p.nonfinite) atomicOr(p.nonfinite, 1);   // rare path: nothing is written when the outputs are finite
  if (p.sampler < 0 && !p.x0 && !p.mean) return;
  __syncthreads();
  const int ts = (int)p.t_idx[b];
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    if (c >= p.C || t >= p.Tn) continue;
    float x0 = tile[tx][i];
    if (p.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    const int64_t o = ((int64_t)b * p.C + c) * p.Tn + t;
    if (p.x0) p.x0[o] = x0;
    const float xv = p.x[o];
    if (p.mean) p.mean[o] = p.tables[TAB_C1 * p.n_steps + ts] * x0 + p.tables[TAB_C2 * p.n_steps + ts] * xv;
    if (p.sampler >= 0) {
      const float nv = p.noise ? p.noise[o] : 0.f;
      p.x_next[o] = p.sampler == 0 ? ddim_update(x0, xv, nv, p.tables, p.n_steps, ts, p.eta)
                                   : ddpm_update(x0, xv, nv, p.tables, p.n_steps, ts);
    }
  }
}

// stand-alone elementwise forms (x, x0, noise all [B, per_sample])
__global__ void ddim_update_kernel(const float* x0, const float* x, const int64_t* t_idx, const float* tab, int ns,
                                   const float* noise, float eta, int64_t per, int64_t total, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  out[i] = ddim_update(x0[i], x[i], noise ? noise[i] : 0.f, tab, ns, t, eta);
}

__global__ void p_sample_update_kernel(const float* mean, const int64_t* t_idx, const float* tab, int ns, const float* noise,
                                       int64_t per, int64_t total, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  const float nz = t != 0 ? 1.f : 0.f;
  out[i] = mean[i] + nz * expf(0.5f * tab[TAB_LOGVAR * ns + t]) * noise[i];
}

// eps = (sqrt(1/abar_t) x_t - x0) / sqrt(1/abar_t - 1)   (gaussian_diffusion.py:347-351)
__global__ void eps_from_xstart_kernel(const float* x, const float* x0, const int64_t* t_idx, const float* tab, int ns, int64_t per,
                                       int64_t total, float* eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  eps[i] = (tab[TAB_SRA * ns + t] * x[i] - x0[i]) / tab[TAB_SRM1 * ns + t];
}

// Pseudo linear multistep update (gaussian_diffusion.py:990-1041).  e0 is the newest eps, e1..e3 older ones.
//   mode 0     : Euler predictor  x0 sqrt(abar_prev) + sqrt(1 - abar_prev) e0       (fed to the model at t-1, no t==0 mask)
//   mode 1..4  : Adams-Bashforth of that order over e0..e{mode-1}
//   mode 5     : improved Euler corrector, eps' = (e1 + e0) / 2   (e1 = eps at t, e0 = eps at the predictor)
__global__ void plms_update_kernel(const float* x, const float* x0, const int64_t* t_idx, const float* tab, int ns, const float* e0,
                                   const float* e1, const float* e2, const float* e3, int mode, int64_t per, int64_t total,
                                   float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  const float abp = tab[TAB_ACPP * ns + t];
  if (mode == 0) {
    out[i] = x0[i] * sqrtf(abp) + sqrtf(1.f - abp) * e0[i];
    return;
  }
  float ep;
  if (mode == 1) ep = e0[i];
  else if (mode == 2) ep = (3.f * e0[i] - e1[i]) / 2.f;
  else if (mode == 3) ep = (23.f * e0[i] - 16.f * e1[i] + 5.f * e2[i]) / 12.f;
  else if (mode == 4) ep = (55.f * e0[i] - 59.f * e1[i] + 37.f * e2[i] - 9.f * e3[i]) / 24.f;
  else ep = (e1[i] + e0[i]) / 2.f;
  const float pred = tab[TAB_SRA * ns + t] * x[i] - tab[TAB_SRM1 * ns + t] * ep;
  const float mean_pred = pred * sqrtf(abp) + sqrtf(1.f - abp) * ep;
  out[i] = t != 0 ? mean_pred : x0[i];
}

// ddim_reverse_sample, eta = 0 (gaussian_diffusion.py:760-795): x0 sqrt(abar_next) + sqrt(1 - abar_next) eps
__global__ void ddim_reverse_kernel(const float* x, const float* x0, const int64_t* t_idx, const float* tab, int ns, int64_t per,
                                    int64_t total, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  const float eps = (tab[TAB_SRA * ns + t] * x[i] - x0[i]) / tab[TAB_SRM1 * ns + t];
  const float abn = tab[TAB_ACPN * ns + t];
  out[i] = x0[i] * sqrtf(abn) + sqrtf(1.f - abn) * eps;
}

__global__ void q_sample_kernel(const float* xs, const int64_t* t_idx, const float* tab, int ns, const float* noise,
                                int64_t per, int64_t total, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = (int)t_idx[i / per];
  out[i] = tab[TAB_SQRT_ACP * ns + t] * xs[i] + tab[TAB_SQRT_1M * ns + t] * noise[i];
}

// dst[i] = src[i] for n float4s: the CFG duplication of the projected input rows (cond half -> uncond half) as an ordinary
// kernel on the launch stream (a hipMemcpyAsync here may be routed to a copy engine; with a second stream active that
// showed up as whole-sample corruption in 1-3 % of forwards on some boxes, scratch/stress2.py)
__global__ void dup_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// gather rows: dst[r] = src[idx[r]]  (fp32 rows of `cols`)
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int rows,
                                   int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  dst[i] = src[(int64_t)idx[r] * cols + c];
}
