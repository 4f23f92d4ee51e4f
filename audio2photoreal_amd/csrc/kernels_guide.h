// Guide transformer (model/guide.py) and residual-VQ decode (model/vqvae.py) kernels -- SURVEY.md section 8 row f2.
//
// The reference's `generate` (model/guide.py:175-222) runs 80 full forwards per sample, each of which re-runs the 13-layer
// 1024-channel `pre_audio` conv stack over all audio tokens and re-encodes the whole token prefix.  Here everything that does
// not depend on the generated tokens is hoisted into a2p_guide_prepare (conv stack as tap-accumulating fp32 MFMA GEMMs, cond
// projection, pooled FiLM vector -> all FiLM scale/shift pairs, rotary + K/V projections of the audio memory for every layer),
// and the token loop is ONE persistent launch: `guide_ar_kernel`, one workgroup per sequence, walks the positions with a
// self-attention K/V cache, runs the 6-layer d=64 stack out of registers/LDS (GEMV-sized work: VALU, no MFMA), and -- in
// sampling mode -- does the softmax, the descending bitonic sort, the nucleus cut (model/guide.py:201-214) and the
// categorical draw (inverse CDF over an injected uniform) in the same kernel.  All arithmetic fp32.
#pragma once
#include "a2p_common.h"

struct GuideLayerW {  // fp32 device pointers of one FiLMTransformerDecoderLayer (transformer_modules.py:127-176)
  const float *ln1_g, *ln1_b, *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;
  const float *ln2_g, *ln2_b, *ca_q_w, *ca_q_b, *ca_out_w, *ca_out_b;
  const float *ln3_g, *ln3_b, *w1, *b1, *w2, *b2;
};

struct GuideArP {
  int d, H, L, ff, V, Vp;       // Vp = V rounded up to a power of two (sort width)
  int Sv, S;                    // valid audio memory rows per sequence, row stride of the K/V caches
  int maxT, n_pos, sc_ld;       // self-attention cache depth, positions to run, row stride of the score buffer
  int mode;                     // 0: teacher forcing (tokens_in, logits_out)   1: sampling (uniforms, tokens_out)
  int start_token;
  float top_p;
  const GuideLayerW* layers;
  const float* tok_emb;         // [V + 1][d]
  const float* fin_w;           // [V][d]
  const float* fin_b;
  const float2* cs;             // rotary table [pos][d/2]
  const float* film;            // [B][L][3][2d]  (scale | shift)
  const float* kc;              // [B][S][L*d]  rotary(memory) Wk^T + bk
  const float* vc;              // [B][S][L*d]  memory Wv^T + bv
  float* sk;                    // [B][L][maxT][d]
  float* sv;
  const int64_t* tokens_in;     // [B][n_pos]
  float* logits_out;            // [B][n_pos][V]
  const float* uniforms;        // [n_pos][B]
  int64_t* tokens_out;          // [B][n_pos]
  float* probs_out;             // optional [n_pos][B][V]: the renormalised sorted nucleus probabilities
};

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over the 256 threads of a block (red: >= 8 floats of LDS); every thread gets the result
__device__ __forceinline__ float guide_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float guide_block_max(float v, float* red) {
  v = wave_max_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// out[r] = act(bias[r] + W[r][0:K] . in), r < N: one thread per output row, float4 weight loads, `in` broadcast from LDS
template <int ACT>
__device__ __forceinline__ void guide_gemv(const float* __restrict__ W, const float* __restrict__ bias, const float* in,
                                           float* out, int N, int K) {
  for (int r = threadIdx.x; r < N; r += 256) {
    const float4* w = reinterpret_cast<const float4*>(W + (int64_t)r * K);
    float acc = bias ? bias[r] : 0.f;
#pragma unroll 8
    for (int k = 0; k < K / 4; ++k) {
      const float4 a = w[k];
      acc += a.x * in[4 * k] + a.y * in[4 * k + 1] + a.z * in[4 * k + 2] + a.w * in[4 * k + 3];
    }
    out[r] = apply_act(acc, ACT);
  }
}

// same for few rows and a long contraction (linear2: N = d, K = ff): wave w takes K-quarter w, partials summed through `part`
__device__ __forceinline__ void guide_gemv_splitk(const float* __restrict__ W, const float* __restrict__ bias, const float* in,
                                                  float* out, float* part, int N, int K) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, kq = K / 4;
  for (int r = lane; r < N; r += 64) {
    const float4* w = reinterpret_cast<const float4*>(W + (int64_t)r * K + wid * kq);
    const float* x = in + wid * kq;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < kq / 4; ++k) {
      const float4 a = w[k];
      acc += a.x * x[4 * k] + a.y * x[4 * k + 1] + a.z * x[4 * k + 2] + a.w * x[4 * k + 3];
    }
    part[wid * N + r] = acc;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < N; r += 256) out[r] = ((part[r] + part[N + r]) + (part[2 * N + r] + part[3 * N + r])) + bias[r];
}

__global__ __launch_bounds__(256) void guide_ar_kernel(const GuideArP p) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int d = p.d, H = p.H, dh = d / H, ff = p.ff, V = p.V, L = p.L;
  float* x = sm;                  // residual row
  float* xh = x + d;              // LayerNorm output
  float* xr = xh + d;             // rotated LayerNorm output
  float* qv = xr + d;             // [2d] q | k of the new position (self attention), q (cross attention)
  float* vv = qv + 2 * d;         // [d]  v of the new position
  float* att = vv + d;            // attention output
  float* tmp = att + d;           // projection output before FiLM
  float* hid = tmp + d;           // [ff]
  float* part = hid + ff;         // [max(4 d, 1024)] partial sums of the split reductions
  float* red = part + (4 * d > 1024 ? 4 * d : 1024);  // [64]
  float* sc = red + 64;           // [H][sc_ld] attention scores / probabilities
  float* lg = sc + H * p.sc_ld;   // [Vp] logits -> probabilities
  int* li = reinterpret_cast<int*>(lg + p.Vp);  // [Vp] token ids carried through the sort
  const float scale = 1.0f / sqrtf((float)dh);

  auto layer_norm = [&](const float* g, const float* be) {
    float s = 0.f;
    for (int e = tid; e < d; e += 256) s += x[e];
    const float mean = guide_block_sum(s, red) / d;
    float q2 = 0.f;
    for (int e = tid; e < d; e += 256) {
      const float dl = x[e] - mean;
      q2 += dl * dl;
    }
    const float rstd = 1.0f / sqrtf(guide_block_sum(q2, red) / d + 1e-5f);
    for (int e = tid; e < d; e += 256) xh[e] = (x[e] - mean) * rstd * g[e] + be[e];
    __syncthreads();
  };
  auto rotate = [&](int pos) {  // rotary_embedding_torch.py:46-66: interleaved pairs over the whole width
    for (int i = tid; i < d / 2; i += 256) {
      const float2 t = p.cs[(int64_t)pos * (d / 2) + i];
      const float a = xh[2 * i], c = xh[2 * i + 1];
      xr[2 * i] = a * t.x - c * t.y;
      xr[2 * i + 1] = c * t.x + a * t.y;
    }
    __syncthreads();
  };
  auto film_residual = [&](const float* fl) {  // x += (scale + 1) * y + shift   (transformer_modules.py:122-124)
    for (int e = tid; e < d; e += 256) x[e] += (fl[e] + 1.0f) * tmp[e] + fl[d + e];
    __syncthreads();
  };
  auto softmax_heads = [&](int n) {  // rows of sc, one wave per head
    for (int h = wid; h < H; h += 4) {
      float* s = sc + h * p.sc_ld;
      float m = -INFINITY;
      for (int k = lane; k < n; k += 64) m = fmaxf(m, s[k]);
      m = wave_max_f(m);
      float z = 0.f;
      for (int k = lane; k < n; k += 64) {
        const float e = expf(s[k] - m);
        s[k] = e;
        z += e;
      }
      z = 1.0f / wave_sum(z);
      for (int k = lane; k < n; k += 64) s[k] *= z;
    }
    __syncthreads();
  };

  int token = p.mode == 0 ? (int)p.tokens_in[(int64_t)b * p.n_pos] : p.start_token;
  for (int pos = 0; pos < p.n_pos; ++pos) {
    if (p.mode == 0) token = (int)p.tokens_in[(int64_t)b * p.n_pos + pos];
    for (int e = tid; e < d; e += 256) x[e] = p.tok_emb[(int64_t)token * d + e];
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      const GuideLayerW w = p.layers[l];
      const float* fl = p.film + ((int64_t)b * L + l) * 3 * 2 * d;
      float* skl = p.sk + ((int64_t)b * L + l) * p.maxT * d;
      float* svl = p.sv + ((int64_t)b * L + l) * p.maxT * d;
      // ---- causal self attention over positions 0..pos (K/V cache) ----
      layer_norm(w.ln1_g, w.ln1_b);
      rotate(pos);
      guide_gemv<ACT_NONE>(w.sa_in_w, w.sa_in_b, xr, qv, 2 * d, d);                         // q | k from the rotated row
      guide_gemv<ACT_NONE>(w.sa_in_w + (int64_t)2 * d * d, w.sa_in_b + 2 * d, xh, vv, d, d);  // v from the plain one
      __syncthreads();
      for (int e = tid; e < d; e += 256) {
        skl[(int64_t)pos * d + e] = qv[d + e];
        svl[(int64_t)pos * d + e] = vv[e];
      }
      __syncthreads();  // the new cache row is visible to the whole workgroup
      for (int i = tid; i < H * (pos + 1); i += 256) {
        const int h = i / (pos + 1), k = i - h * (pos + 1);
        const float* kr = skl + (int64_t)k * d + h * dh;
        float s = 0.f;
        for (int j = 0; j < dh; ++j) s += qv[h * dh + j] * kr[j];
        sc[h * p.sc_ld + k] = s * scale;
      }
      __syncthreads();
      softmax_heads(pos + 1);
      for (int e = tid; e < d; e += 256) {
        const float* pr = sc + (e / dh) * p.sc_ld;
        float a = 0.f;
        for (int k = 0; k <= pos; ++k) a += pr[k] * svl[(int64_t)k * d + e];
        att[e] = a;
      }
      __syncthreads();
      guide_gemv<ACT_NONE>(w.sa_out_w, w.sa_out_b, att, tmp, d, d);
      __syncthreads();
      film_residual(fl);
      // ---- cross attention over the Sv cached audio-memory rows ----
      layer_norm(w.ln2_g, w.ln2_b);
      rotate(pos);
      guide_gemv<ACT_NONE>(w.ca_q_w, w.ca_q_b, xr, qv, d, d);
      __syncthreads();
      const float* kcl = p.kc + (int64_t)b * p.S * L * d + (int64_t)l * d;
      const float* vcl = p.vc + (int64_t)b * p.S * L * d + (int64_t)l * d;
      for (int k = tid; k < p.Sv; k += 256) {
        const float4* kr = reinterpret_cast<const float4*>(kcl + (int64_t)k * L * d);
        for (int h = 0; h < H; ++h) {
          float s = 0.f;
          for (int j = 0; j < dh / 4; ++j) {
            const float4 a = kr[h * (dh / 4) + j];
            const float* qq = qv + h * dh + 4 * j;
            s += a.x * qq[0] + a.y * qq[1] + a.z * qq[2] + a.w * qq[3];
          }
          sc[h * p.sc_ld + k] = s * scale;
        }
      }
      __syncthreads();
      softmax_heads(p.Sv);
      {  // P.V: d/4 threads read one V row as float4s, 256/(d/4) key groups run in parallel, partials reduced through LDS
        const int tpr = d / 4, kg = 256 / tpr, e4 = tid % tpr, grp = tid / tpr;
        if (grp < kg) {
          const float* pr = sc + ((4 * e4) / dh) * p.sc_ld;
          float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
          for (int k = grp; k < p.Sv; k += kg) {
            const float4 v = *reinterpret_cast<const float4*>(vcl + (int64_t)k * L * d + 4 * e4);
            const float w8 = pr[k];
            a.x += w8 * v.x; a.y += w8 * v.y; a.z += w8 * v.z; a.w += w8 * v.w;
          }
          *reinterpret_cast<float4*>(part + grp * d + 4 * e4) = a;
        }
        __syncthreads();
        for (int e = tid; e < d; e += 256) {
          float a = 0.f;
          for (int q8 = 0; q8 < kg; ++q8) a += part[q8 * d + e];
          att[e] = a;
        }
        __syncthreads();
      }
      guide_gemv<ACT_NONE>(w.ca_out_w, w.ca_out_b, att, tmp, d, d);
      __syncthreads();
      film_residual(fl + 2 * d);
      // ---- feed forward ----
      layer_norm(w.ln3_g, w.ln3_b);
      guide_gemv<ACT_GELU>(w.w1, w.b1, xh, hid, ff, d);
      __syncthreads();
      guide_gemv_splitk(w.w2, w.b2, hid, tmp, part, d, ff);
      __syncthreads();
      film_residual(fl + 4 * d);
    }
    guide_gemv<ACT_NONE>(p.fin_w, p.fin_b, x, lg, V, d);
    __syncthreads();
    if (p.mode == 0) {
      for (int v = tid; v < V; v += 256) p.logits_out[((int64_t)b * p.n_pos + pos) * V + v] = lg[v];
      __syncthreads();
      continue;
    }
    // ---- softmax -> descending sort -> nucleus -> categorical draw (model/guide.py:200-217) ----
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, lg[v]);
    m = guide_block_max(m, red);
    float z = 0.f;
    for (int v = tid; v < V; v += 256) z += expf(lg[v] - m);
    z = guide_block_sum(z, red);
    for (int v = tid; v < p.Vp; v += 256) {
      lg[v] = v < V ? expf(lg[v] - m) / z : -1.0f;  // padding sorts to the end
      li[v] = v;
    }
    __syncthreads();
    for (int k = 2; k <= p.Vp; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < p.Vp; i += 256) {
          const int o = i ^ j;
          if (o > i) {
            const bool desc = (i & k) == 0;  // overall descending
            const float a = lg[i], c = lg[o];
            if (desc ? a < c : a > c) {
              lg[i] = c; lg[o] = a;
              const int t = li[i]; li[i] = li[o]; li[o] = t;
            }
          }
        }
        __syncthreads();
      }
    if (tid == 0) {  // sequential like torch.cumsum: the nucleus keeps entry i while the cumulative mass BEFORE it is < top_p
      float c = 0.f, kept = 0.f;
      int n = 0;
      for (int i = 0; i < V; ++i) {
        if (i > 0 && !(c < p.top_p)) break;
        c += lg[i];
        kept += lg[i];
        ++n;
      }
      const float u = p.uniforms[(int64_t)pos * gridDim.x + b];
      int pick = 0;
      float c2 = 0.f;
      for (int i = 0; i < n; ++i) {
        c2 += lg[i] / kept;
        if (c2 > u) { pick = i; break; }
      }
      red[8] = kept;
      reinterpret_cast<int*>(red)[9] = n;
      reinterpret_cast<int*>(red)[10] = li[pick];
    }
    __syncthreads();
    token = reinterpret_cast<int*>(red)[10];
    if (p.probs_out) {
      const float kept = red[8];
      const int n = reinterpret_cast<int*>(red)[9];
      for (int v = tid; v < V; v += 256) p.probs_out[((int64_t)pos * gridDim.x + b) * V + v] = v < n ? lg[v] / kept : 0.f;
    }
    if (tid == 0) p.tokens_out[(int64_t)b * p.n_pos + pos] = token;
    __syncthreads();
  }
}

// ---- hoisted conditioning (model/guide.py:150-169) ------------------------------------------------------------------------
// Conv1d weight [Co][Ci][taps] -> [tap][Co][Ci] (the tap-accumulating GEMM's layout)
__global__ void guide_conv_repack_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int taps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Co * Ci * taps) return;
  const int t = (int)(i % taps);
  const int64_t oc = i / taps;  // co * Ci + ci
  dst[(int64_t)t * Co * Ci + oc] = src[i];
}

// mean over the Sv valid rows of each sequence (row stride S): src [B][S][d] -> dst [B][d]
__global__ void guide_mean_kernel(const float* __restrict__ src, float* __restrict__ dst, int S, int Sv, int d) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const float* q = src + (int64_t)b * S * d + c;
  float s = 0.f;
  for (int i = 0; i < Sv; ++i) s += q[(int64_t)i * d];
  dst[(int64_t)b * d + c] = s / (float)Sv;
}

struct GuideHiddenP {
  const float* pooled;                      // [B][d]
  const float *ln_g, *ln_b, *w1, *b1, *w3, *b3;  // non_attn_cond_projection.{0,1,3}
  const float* null_hidden;                 // [d], used when drop
  int d, drop;
  float* hidden;                            // [B][d]
  float* mish_hidden;                       // [B][d]  input of every DenseFiLM (transformer_modules.py:111-113)
};
// LayerNorm -> Linear -> SiLU -> Linear of the pooled tokens, one workgroup per sequence (d <= 512)
__global__ __launch_bounds__(256) void guide_hidden_kernel(const GuideHiddenP p) {
  __shared__ float a[512], h[512], red[64];
  const int b = blockIdx.x, tid = threadIdx.x, d = p.d;
  const float* x = p.pooled + (int64_t)b * d;
  float s = 0.f;
  for (int e = tid; e < d; e += 256) s += x[e];
  const float mean = guide_block_sum(s, red) / d;
  float q2 = 0.f;
  for (int e = tid; e < d; e += 256) {
    const float dl = x[e] - mean;
    q2 += dl * dl;
  }
  const float rstd = 1.0f / sqrtf(guide_block_sum(q2, red) / d + 1e-5f);
  for (int e = tid; e < d; e += 256) a[e] = (x[e] - mean) * rstd * p.ln_g[e] + p.ln_b[e];
  __syncthreads();
  guide_gemv<ACT_SILU>(p.w1, p.b1, a, h, d, d);
  __syncthreads();
  guide_gemv<ACT_NONE>(p.w3, p.b3, h, a, d, d);
  __syncthreads();
  for (int e = tid; e < d; e += 256) {
    const float v = p.drop ? p.null_hidden[e] : a[e];
    p.hidden[(int64_t)b * d + e] = v;
    p.mish_hidden[(int64_t)b * d + e] = act_mish(v);
  }
}

// norm_cond LayerNorm of the audio tokens and their rotated copy (position = row inside the sequence): one wave per row.
// src rows come from the cond projection ([B][S][d]) or, when null_embed != NULL, from null_cond_embed[row] for every sequence.
__global__ __launch_bounds__(256) void guide_memory_kernel(const float* __restrict__ src, const float* __restrict__ null_embed,
                                                          const float* __restrict__ g, const float* __restrict__ be,
                                                          const float2* __restrict__ cs, float* __restrict__ mem,
                                                          float* __restrict__ memr, float* __restrict__ ct_out, int S, int Sv, int d,
                                                          int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int pos = row % S;
  const float* x = null_embed ? null_embed + (int64_t)(pos < Sv ? pos : 0) * d : src + (int64_t)row * d;
  float v[8];  // d <= 512
  float s = 0.f;
  for (int i = 0; i < d / 64; ++i) {
    v[i] = x[lane + 64 * i];
    s += v[i];
  }
  const float mean = wave_sum(s) / d;
  float q2 = 0.f;
  for (int i = 0; i < d / 64; ++i) q2 += (v[i] - mean) * (v[i] - mean);
  const float rstd = 1.0f / sqrtf(wave_sum(q2) / d + 1e-5f);
  for (int i = 0; i < d / 64; ++i) {
    const int e = lane + 64 * i;
    if (ct_out) ct_out[(int64_t)row * d + e] = v[i];
    const float n = (v[i] - mean) * rstd * g[e] + be[e];
    mem[(int64_t)row * d + e] = n;
    const float other = __shfl_xor(n, 1, 64);  // partner of the interleaved pair (e ^ 1 sits in lane ^ 1)
    const float2 t = cs[(int64_t)pos * (d / 2) + (e >> 1)];
    memr[(int64_t)row * d + e] = (e & 1) ? n * t.x + other * t.y : n * t.x - other * t.y;
  }
}

// ---- residual VQ decode (model/vqvae.py:508-521, 381-392, 452-463): one workgroup per sequence ---------------------------
struct VqDecodeP {
  const int64_t* q;            // [B][T][depth]
  const float* codebook[8];    // depth x [categories][e]
  const float* cw[5];          // dec.{0,2,4,6}: [e][e][2], dec.8: [nv][e][1]
  const float* cb[5];
  int T, depth, e, nv;
  float* out;                  // [B][T][nv]
};
__global__ __launch_bounds__(256) void vq_decode_kernel(const VqDecodeP p) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, T = p.T, e = p.e, R = T + 7;  // 7 zero rows of left padding (receptive field 8)
  float* cur = sm;            // [R][e]
  float* nxt = cur + R * e;   // [R][e]
  for (int i = tid; i < R * e; i += 256) {
    const int r = i / e, c = i - r * e;
    float v = 0.f;
    if (r >= 7)
      for (int k = 0; k < p.depth; ++k) v += p.codebook[k][p.q[((int64_t)b * T + (r - 7)) * p.depth + k] * e + c];
    cur[i] = v;
  }
  __syncthreads();
  const int dil[4] = {1, 2, 3, 1};
  int first = 0;  // rows [first, R) of `cur` are valid
  for (int l = 0; l < 4; ++l) {
    const int nf = first + dil[l];  // a valid (unpadded) k=2 conv shortens the front by its dilation
    for (int i = tid; i < (R - nf) * e; i += 256) {
      const int r = nf + i / e, co = i % e;
      const float* w = p.cw[l] + (int64_t)co * e * 2;
      float acc = p.cb[l][co];
      for (int ci = 0; ci < e; ++ci) acc += w[2 * ci] * cur[(r - dil[l]) * e + ci] + w[2 * ci + 1] * cur[r * e + ci];
      nxt[r * e + co] = act_lrelu02(acc);
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
    first = nf;
  }
  for (int i = tid; i < T * p.nv; i += 256) {  // first == 7: rows 7.. are the T outputs
    const int t = i / p.nv, co = i - t * p.nv;
    const float* w = p.cw[4] + (int64_t)co * e;
    float acc = p.cb[4][co];
    for (int ci = 0; ci < e; ++ci) acc += w[ci] * cur[(7 + t) * e + ci];
    p.out[((int64_t)b * T + t) * p.nv + co] = acc;
  }
}
