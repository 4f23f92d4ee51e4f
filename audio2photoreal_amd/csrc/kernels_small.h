// Small-problem GEMMs of the decoder layer (16-bit modes, forwards of fewer than 1100 rows: BASELINE configs[0] is 480 rows).
//
// A 480-row forward does not fill the chip with row panels (10 panels of 48 rows; a chain workgroup's time does not depend on how
// many panels there are) and as per-operation launches it is 111 dependent kernels of 4-12 us, none of which is long enough to
// amortise its own latency: a 64x128-tile GEMM walks K = 512 as 8 dependent tile loads.  This kernel is built for that regime:
//   * the WHOLE contraction is resident: a workgroup's [32 x K] A panel and [BN x K] weight tile are requested in one burst
//     (global_load_lds, 16 B per lane, XOR-swizzled on the source address) and waited for once -- one memory round trip per launch;
//   * LayerNorm (+ rotary) is FUSED into the A load: the panel is normalised from the fp32 residual rows while the weight tile is
//     in flight (every column tile of a row block repeats the 32-row LayerNorm: 32 x 512 values, cheaper than a launch); the
//     25 LayerNorm launches of a step disappear;
//   * small tiles (32 x 64, or 32 x 32 for K = 1024) give 120-360 workgroups per GEMM at 480 rows;
//   * [Q|K] (rotated panel) and V (plain panel, transposed store) of the self attention are ONE launch.
// 8 launches per decoder layer instead of 12.  Arithmetic per output element as in gemm_kernel / ln_rope_kernel (same MFMA shape,
// k ascending, fp32 accumulate, same epilogue formulas).
#pragma once
#include "a2p_common.h"

enum { SMALL_STORE = 0, SMALL_FILM_RES = 1 };

struct SmallP {
  // A operand: fp32 residual rows (LayerNorm prologue) or 16-bit rows (plain)
  const float* x;          // [M][K], row stride K                      (PRO = 1)
  const h16_t* a;          // [M][lda]                                  (PRO = 0)
  int64_t lda;
  const float* gamma;      // LayerNorm affine (PRO = 1)
  const float* beta;
  const float2* cs;        // rotary table [pos][K/2] (cos, sin)
  int rows_per_seq;
  int n_rope;              // column tiles starting below n_rope take the ROTATED panel, the others the plain one
  const h16_t* W;          // [N][K]
  const float* bias;       // [N]
  int M, N;
  // SMALL_STORE: columns [0, n_store) row-major into out, columns >= n_store transposed (V^T) into out_t
  h16_t* out;
  int64_t ldo;
  int n_store;
  h16_t* out_t;            // out_t[seq * t_seq_stride + (n - n_store) * ld_t + (m - seq * rows_per_seq)]
  int64_t ld_t, t_seq_stride;
  int gelu;                // activation of the stored values (linear1)
  // SMALL_FILM_RES: resid[m][n] += (film_scale + 1) * (acc + bias) + film_shift   (fp32, in place)
  float* resid;
  const float* film;       // scale at film[seq * film_seq_stride + n], shift at + film_shift_off
  int64_t film_seq_stride;
  int film_shift_off;
};

// BM: 32 rows.  (64-row tiles for the wide LayerNorm-prologue GEMMs -- one round of workgroups instead of 1.4 at 480 rows, half the
// weight traffic -- were measured and lose: 14.8 us per launch against 12.5: the prologue of a workgroup is a serial chain of
// latencies, twice as long with twice the rows.)
// LayerNorm-prologue launches (PRO = 1) keep their weight tile in REGISTERS (WREG): a wave's 16 columns x K = 512 are 16 x 16 bytes
// per lane, requested straight from global memory behind the residual rows.  Only the A panel is in LDS then (32 KiB instead of
// 96), so two workgroups share a CU and the 360 workgroups of the [Q|K|V] launch at 480 rows are one round instead of 1.4, each
// hiding the other's LayerNorm round trips.
template <int K, int BN, int PRO, int EPI, int BM = 32>
__global__ __launch_bounds__(256, PRO == 1 ? 2 : 1) void small_gemm_kernel(const SmallP p) {
  constexpr int CPR = K / 8;                          // 16-byte chunks per row
  constexpr int WR = BN == 64 ? 1 : 2;                // wave grid: WR row groups x (4 / WR) column groups of 16 columns
  constexpr int MT = BM / 16 / WR;                    // 16-row tiles per wave
  constexpr int KC = K / 32;                          // MFMA k-chunks
  constexpr bool WREG = PRO == 1;
  static_assert((BN == 64 || BN == 32) && (K == 512 || K == 1024) && (BM == 32 || BM == 64), "tile shapes");
  static_assert(!WREG || (BN == 64 && K == 512), "register-resident weight tile: one 16-column slice per wave");
  __shared__ __attribute__((aligned(16))) h16_t smem[(BM + (WREG ? 0 : BN)) * K];
  h16_t* const As = smem;
  h16_t* const Ws = smem + BM * K;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // ---- weight tile: BN rows of K values, one burst.  LDS chunk position q of row r holds global chunk q ^ (r & 15) ------------
  if constexpr (!WREG) {
    constexpr int RPI = 64 / CPR > 0 ? 64 / CPR : 1;   // rows per wave instruction (K = 512: 1)
    constexpr int IPR = CPR / 64 > 0 ? CPR / 64 : 1;   // instructions per row (K = 1024: 2)
    static_assert(RPI == 1, "K >= 512");
    for (int r = wid; r < BN; r += 4) {
      int gn = n0 + r;
      gn = gn < p.N ? gn : p.N - 1;
#pragma unroll
      for (int i = 0; i < IPR; ++i) {
        const int q = i * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + (int64_t)gn * K + ((q ^ (r & 15)) << 3)),
                                         (__attribute__((address_space(3))) void*)(Ws + r * K + i * 512), 16, 0, 0);
      }
    }
  }
  [[maybe_unused]] h16x8 wreg[WREG ? KC : 1];
  // ---- A panel ------------------------------------------------------------------------------------------------------------
  if constexpr (PRO == 0) {
    constexpr int IPR = CPR / 64;
    for (int r = wid; r < BM; r += 4) {
      int gm = m0 + r;
      gm = gm < p.M ? gm : p.M - 1;
#pragma unroll
      for (int i = 0; i < IPR; ++i) {
        const int q = i * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.a + (int64_t)gm * p.lda + ((q ^ (r & 15)) << 3)),
                                         (__attribute__((address_space(3))) void*)(As + r * K + i * 512), 16, 0, 0);
      }
    }
  } else {
    static_assert(PRO == 0 || K == 512, "the LayerNorm prologue is written for d_model = 512");
    // LayerNorm (+ rotary) of rows wid*8 .. +7: a lane holds 8 consecutive values of a row (ln_rope_kernel's arithmetic)
    const bool rope = n0 < p.n_rope;   // workgroup-uniform
    const int e0 = lane * 8;
    float4 ga0 = *reinterpret_cast<const float4*>(p.gamma + e0), ga1 = *reinterpret_cast<const float4*>(p.gamma + e0 + 4);
    float4 be0 = *reinterpret_cast<const float4*>(p.beta + e0), be1 = *reinterpret_cast<const float4*>(p.beta + e0 + 4);
    const float gam[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
    const float bet[8] = {be0.x, be0.y, be0.z, be0.w, be1.x, be1.y, be1.z, be1.w};
#pragma unroll 1
    for (int rb = 0; rb < BM; rb += 32) {   // 8 rows per wave and batch
    float v[8][8];
    float4 rc[8][2];   // the rows' rotary entries (cos, sin) x 4 pairs, requested with the rows: one round trip, not two
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {   // all loads of the wave's rows first
      int gm = m0 + rb + wid * 8 + rr;
      gm = gm < p.M ? gm : p.M - 1;
      const float4 t0 = *reinterpret_cast<const float4*>(p.x + (int64_t)gm * K + e0);
      const float4 t1 = *reinterpret_cast<const float4*>(p.x + (int64_t)gm * K + e0 + 4);
      v[rr][0] = t0.x; v[rr][1] = t0.y; v[rr][2] = t0.z; v[rr][3] = t0.w;
      v[rr][4] = t1.x; v[rr][5] = t1.y; v[rr][6] = t1.z; v[rr][7] = t1.w;
      if (rope) {
        const float4* c4 = reinterpret_cast<const float4*>(p.cs + (int64_t)(gm % p.rows_per_seq) * (K / 2) + e0 / 2);
        rc[rr][0] = c4[0];
        rc[rr][1] = c4[1];
      }
    }
    if constexpr (WREG) {   // the wave's weight slice, behind the rows (vmcnt is in order: the LayerNorm does not wait for it)
      if (rb == 0) {
        int gn = n0 + (WR == 1 ? wid : (wid & 1)) * 16 + l15;
        gn = gn < p.N ? gn : p.N - 1;
        const h16_t* wg = p.W + (int64_t)gn * K + g * 8;
#pragma unroll
        for (int c = 0; c < KC; ++c) wreg[c] = *reinterpret_cast<const h16x8*>(wg + c * 32);
      }
    }
    // The two reductions of the 8 rows are independent chains of VALU cross-lane steps (DPP + permlane swaps, a2p_common.h): as
    // __shfl_xor butterflies (ds_bpermute: an LDS round trip of ~120 cycles per step) done row by row they were 8 us of a 14 us launch
    auto wave_sum8 = [&](float(&t)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) t[rr] = wave_sum_valu(t[rr]);
    };
    float red[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[rr][i];
      red[rr] = s;
    }
    wave_sum8(red);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const float mean = red[rr] * (1.0f / K);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[rr][i] -= mean;
        q += v[rr][i] * v[rr][i];
      }
      red[rr] = q;
    }
    wave_sum8(red);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = rb + wid * 8 + rr;
      const float rstd = 1.0f / sqrtf(red[rr] * (1.0f / K) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[rr][i] = v[rr][i] * rstd * gam[i] + bet[i];
      h16x8 o;
      if (rope) {
        const float cs8[8] = {rc[rr][0].x, rc[rr][0].y, rc[rr][0].z, rc[rr][0].w, rc[rr][1].x, rc[rr][1].y, rc[rr][1].z, rc[rr][1].w};
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const float tc = cs8[i], ts = cs8[i + 1];
          o[i] = (h16_t)(v[rr][i] * tc - v[rr][i + 1] * ts);
          o[i + 1] = (h16_t)(v[rr][i + 1] * tc + v[rr][i] * ts);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (h16_t)v[rr][i];
      }
      *reinterpret_cast<h16x8*>(As + r * K + ((lane ^ (r & 15)) << 3)) = o;
    }
    }
  }
  // ---- MFMA: wave (wr, wc) owns rows wr*16*MT .. and columns wc*16 .. +15 of the tile ------------------------------------------
  const int wr = WR == 1 ? 0 : (wid >> 1), wc = WR == 1 ? wid : (wid & 1);
  // FiLM / residual operands of the epilogue are requested NOW, with the tiles: one memory round trip per launch instead of two
  // (each output element belongs to exactly one lane of one workgroup, so the residual read cannot race with another writer)
  [[maybe_unused]] float4 ex[MT], esc[MT], esh[MT];
  if constexpr (EPI == SMALL_FILM_RES) {
    const int n = n0 + wc * 16 + g * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = m0 + (wr * MT + mt) * 16 + l15;
      m = m < p.M ? m : p.M - 1;
      const int nn = n < p.N ? n : 0;
      ex[mt] = *reinterpret_cast<const float4*>(p.resid + (int64_t)m * p.N + nn);
      const float* fp = p.film + (int64_t)(m / p.rows_per_seq) * p.film_seq_stride + nn;
      esc[mt] = *reinterpret_cast<const float4*>(fp);
      esh[mt] = *reinterpret_cast<const float4*>(fp + p.film_shift_off);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const bool transposed = EPI == SMALL_STORE && p.out_t != nullptr && n0 >= p.n_store;   // workgroup-uniform
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const h16_t* arow = As + (wr * 16 * MT + l15) * K;
  const h16_t* wrow = Ws + (wc * 16 + l15) * K;
  // fragments of 8 k-chunks are requested ahead of the MFMAs that consume them: the kernel is a chain of latencies (launch, one
  // memory round trip, this loop, the stores), an un-pipelined loop pays one LDS round trip per k-chunk
  constexpr int CB = 8;
  auto ksteps = [&](int c0) __attribute__((always_inline)) {
    h16x8 wf[CB], af[CB][MT];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int pos = (((c0 + i) * 4 + g) ^ l15) << 3;
      if constexpr (WREG) wf[i] = wreg[c0 + i];
      else wf[i] = *reinterpret_cast<const h16x8*>(wrow + pos);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[i][mt] = *reinterpret_cast<const h16x8*>(arow + mt * 16 * K + pos);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (transposed) acc[mt] = A2P_MFMA16(af[i][mt], wf[i], acc[mt]);   // D = C: lane holds rows g*4 + r of column l15
        else acc[mt] = A2P_MFMA16(wf[i], af[i][mt], acc[mt]);              // D = C^T: lane holds columns g*4 + r of row l15
      }
  };
  if constexpr (WREG) {   // register-resident weights: compile-time chunk indices
#pragma unroll
    for (int c0 = 0; c0 < KC; c0 += CB) ksteps(c0);
  } else {
#pragma unroll 1
    for (int c0 = 0; c0 < KC; c0 += CB) ksteps(c0);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  if (transposed) {
    const int n = n0 + wc * 16 + l15;
    const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + (wr * MT + mt) * 16 + g * 4;
      if (m >= p.M || n >= p.N) continue;
      const f32x4 v = acc[mt] + b;
      const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
      const int sq = m / p.rows_per_seq;
      h16_t* dst = p.out_t + (int64_t)sq * p.t_seq_stride + (int64_t)(n - p.n_store) * p.ld_t;
      if ((p.rows_per_seq & 3) == 0) {
        *reinterpret_cast<h16x4*>(dst + (m - sq * p.rows_per_seq)) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = m + r;
          if (mm >= p.M) break;
          const int s2 = mm / p.rows_per_seq;
          p.out_t[(int64_t)s2 * p.t_seq_stride + (int64_t)(n - p.n_store) * p.ld_t + (mm - s2 * p.rows_per_seq)] = o[r];
        }
      }
    }
    return;
  }
  const int n = n0 + wc * 16 + g * 4;
  if (n >= p.N) return;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + (wr * MT + mt) * 16 + l15;
    if (m >= p.M) continue;
    f32x4 v = acc[mt];
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    if constexpr (EPI == SMALL_FILM_RES) {
      float4* xp = reinterpret_cast<float4*>(p.resid + (int64_t)m * p.N + n);
      float4 x = ex[mt];
      const float4 sc = esc[mt], sh = esh[mt];
      x.x += (sc.x + 1.0f) * v[0] + sh.x;
      x.y += (sc.y + 1.0f) * v[1] + sh.y;
      x.z += (sc.z + 1.0f) * v[2] + sh.z;
      x.w += (sc.w + 1.0f) * v[3] + sh.w;
      *xp = x;
    } else {
      if (p.gelu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_gelu_fast(v[r]);
      }
      *reinterpret_cast<h16x4*>(p.out + (int64_t)m * p.ldo + n) = h16x4{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
    }
  }
}
