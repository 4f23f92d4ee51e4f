// Row-panel chain kernels, second generation (round 3): the same three chains as kernels_chain.h (PRE / MID / POST of
// FiLMTransformerDecoderLayer.forward, transformer_modules.py:178-267) and the same bits, restructured around what
// scratch/tall_probe.hip measured (profiles/r03_tall_probe.txt):
//
//   * WEIGHTS GO STRAIGHT FROM L2 INTO VGPRs.  Every wave owns 16 of a tile's 128 output columns for ALL rows of the panel, so
//     a weight fragment has exactly one consumer: staging it through LDS (kernels_chain.h: LDS-DMA into a wave-private ring,
//     then ds_read_b128) costs an LDS write, an LDS read and 80 KiB of ring for nothing.  The stream is packed so that one
//     global_load_dwordx4 of a wave IS one MFMA operand (64 lanes x 16 B contiguous = 1 KiB); a 4-stage register ring (32 VGPRs)
//     runs ahead across tiles, GEMMs and epilogues.  The probe: 48 rows 326 -> 287 cycles per 16 KiB stage (the 64 B/clk TA path
//     is the floor there), 64 / 80 / 96 rows 263 / 315 / 380 cycles = the MFMA floor itself (6.8 -> 3.9-4.1 cycles per row).
//   * TALL PANELS.  Without the ring the LDS holds an 80- or 96-row panel (B=32: 480 workgroups of 80 rows = two rounds of the
//     256 CUs instead of three rounds of 48/64-row panels), and 80 FLOP per streamed weight byte make the stage MFMA-bound
//     (320 cycles of MFMA against 256 of weight traffic).
//   * THE RESIDUAL ROWS ARE NOT REGISTER-RESIDENT ACROSS THE GEMMS.  They are read when an epilogue needs them (FiLM + residual
//     folds them into the accumulators IN PLACE: the accumulators become the residual rows), normalised, written to the panel and
//     stored back ("parked") before the next GEMM group; the feed-forward block therefore runs with 80 accumulator registers
//     instead of 160, which is what lets two 256-register waves per SIMD carry 80-row panels.  Cost: one extra 2 KiB store + load
//     per row and POST kernel, through L2.
//   * Fragment reads of the A panel are software-pipelined half a stage ahead (tile GEMMs) or one k-step ahead (k-major group
//     GEMMs) and pinned in front of the MFMAs that cover their latency.
//
// 8 waves per workgroup (two per SIMD, 256 registers each), one workgroup per CU.  Column ownership, accumulation order per
// output element, the 8-partial LayerNorm tree and the epilogue arithmetic are those of kernels_chain.h, so the two generations
// are bit-identical (tests/test_hip_round3.py) and the host may pick either.
#pragma once
#include "kernels_chain.h"

#pragma clang fp contract(off)

#ifndef C2_ABL
#define C2_ABL 0   // compile-time ablation bits for register-pressure hunts (scratch only): 1 out_proj, 2 film_res, 4 LayerNorm, 8 gemm_store
#endif
#define CHAIN2_PF 4
// phase boundary: nothing is scheduled across (each phase of a chain is its own scheduling region -- in one region hipcc
// interleaves the address arithmetic and loads of later phases with the GEMM in front of them and runs out of registers)
#define C2_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)   // weight stages in flight per wave (register ring); every GEMM of a chain consumes a multiple of it

template <int D, int MT>
struct Chain2Lds {
  static constexpr int BM = 16 * MT, AUX_F = 2560;
  static constexpr int ELEMS = BM * D + BM * 128 + 32 * BM + 2 * AUX_F;   // 16-bit elements: panelA, panelH, LN partials, aux
};

// Repack of one 16 KiB stage for the direct path: [wave 0..7][k-chunk 0..1][lane 0..63][8 k-values] -- lane (l15, g) of wave w
// gets W[col(w, l15)][k0 + (kk*4 + g)*8 .. +8], the B operand of v_mfma_f32_16x16x32 for k-chunk kk.  Column ownership as in
// chain_pack_kernel (8 waves: 32-column group w >> 1, sub-tile w & 1, paired map for stored tiles).
__global__ __launch_bounds__(256) void chain2_pack_kernel(const ChainPackDesc* __restrict__ descs, h16_t* __restrict__ dst) {
  const ChainPackDesc d = descs[blockIdx.x];
  uint4* out = reinterpret_cast<uint4*>(dst + (int64_t)blockIdx.x * CHAIN_STAGE_ELEMS);
  for (int q = threadIdx.x; q < 1024; q += 256) {
    const int w = q >> 7, kk = (q >> 6) & 1, lane = q & 63, i = lane & 15, g = lane >> 4;
    const int w4 = w >> 1, J = w & 1, tile = d.row0 >> 7;
    const int row = d.omap ? (tile >> 1) * 256 + w4 * 64 + J * 32 + (tile & 1) * 16 + i : d.row0 + w4 * 32 + (i >> 2) * 8 + J * 4 + (i & 3);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < d.nrows) v = *reinterpret_cast<const uint4*>(d.W + (int64_t)row * d.ldw + d.k0 + (kk * 4 + g) * 8);
    out[q] = v;
  }
}

template <int D, int MT, int MODE>
__device__ __forceinline__ void chain2_body(const ChainP& p, h16_t* const smem, const int m0) {
  constexpr int NW = 8, CW = 16, BM = 16 * MT, CPR = D / 8, NT = D / 128, KS = D / 64, FT = 8, HLD = 128, AUX_F = 2560, PF = CHAIN2_PF;
  static_assert(KS % PF == 0 && (2 * NT) % PF == 0, "every GEMM must consume a multiple of the register ring");
  h16_t* const panelA = smem;
  h16_t* const panelH = panelA + BM * D;
  float* const red = reinterpret_cast<float*>(panelH + BM * HLD);   // [2][8][BM] LayerNorm partial sums, one per wave
  float* const aux = red + 16 * BM;                                   // [AUX_F] per-tile biases
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int W4 = wid >> 1, J0 = wid & 1;
#if defined(C2_STAMPS) || defined(A2P_STAMPS)   // scratch/chain2_bench.hip, scratch/phase_probe.py: 100 MHz phase stamps of workgroups 0 and 101 into p.fin_out
  auto stamp = [&](int i) __attribute__((always_inline)) {
    const int cb = (int)blockIdx.x - p.n_pf;
    if (tid == 0 && (cb == 0 || cb == 101)) reinterpret_cast<unsigned long long*>(p.fin_out)[(cb ? 32 : 0) + i] = wall_clock64();
  };
#else
  auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
  stamp(0);

  // ---- weight stream: a register ring of PF stages, 2 fragments (k-chunks) per stage ------------------------------------
  // Ordinary (compiler-visible) loads: hipcc then places the s_waitcnt vmcnt(N) itself and knows that a ring register is not
  // valid before it -- as inline asm the loads worked until the register allocator copied a ring register across a loop
  // back-edge BEFORE the hand-written wait (a copy of a register whose load is still in flight copies garbage).
  uint32_t woff = (uint32_t)(wid * 128 + lane) * 16;   // byte offset of this lane's 16 bytes of k-chunk 0 of the next stage to load
  h16x8 wr[PF][2];
  auto w_issue = [&](int slot) __attribute__((always_inline)) {
    // default cache policy: every workgroup of the launch walks the same stream, L2 serves all but the first
    const char* q = reinterpret_cast<const char*>(p.stream) + woff;   // uniform base + 32-bit offset: SGPR-base addressing
    wr[slot][0] = *reinterpret_cast<const h16x8*>(q);
    wr[slot][1] = *reinterpret_cast<const h16x8*>(q + 1024);
    woff += 16384;            // the host pads CHAIN_STREAM_PAD stages behind the last one
  };
  auto w_wait = [&](int) __attribute__((always_inline)) {};

  // ---- helpers ----------------------------------------------------------------------------------------------------------
  auto col_of = [&](int t) __attribute__((always_inline)) { return t * 128 + W4 * 32 + g * 8 + J0 * 4; };
  auto obase = [&](int t) __attribute__((always_inline)) { return (t >> 1) * 256 + W4 * 64 + J0 * 32 + (t & 1) * 16; };
  // Residual-row addressing with 32-bit element offsets (a forward holds < 2^32 bytes of rows): offset of this lane's 4 columns of
  // tile t of row m = x_rbase(m, layout) + t * x_tstride(layout).  Tiled layout (ChainP::x_in_tiled): per 16-row block the D/16
  // chunks (tile t, 32-column group W4, half J0) of 1 KiB each, chunk = [g][row & 15][4 floats]; row-major: m * D + col_of(t).
  // One select per row instead of a 64-bit address expression (and a branch) per (row, tile).
  auto x_rbase = [&](int m, int tiled) __attribute__((always_inline)) -> uint32_t {
    const uint32_t a = (uint32_t)(((m >> 4) * (D / 16) + W4 * 2 + J0) * 256 + (g * 16 + (m & 15)) * 4);
    const uint32_t b = (uint32_t)(m * D + W4 * 32 + g * 8 + J0 * 4);
    return tiled ? a : b;
  };
  auto x_tstride = [&](int tiled) __attribute__((always_inline)) -> uint32_t { return tiled ? 2048u : 128u; };
  // global accesses as (uniform base pointer) + (32-bit BYTE offset): the form global_load/store take as SGPR base + VGPR offset.
  // Element offsets scaled by the compiler become 64-bit address arithmetic per access (two registers and a v_lshl_add_u64 each).
  auto ld4 = [&](const float* base, uint32_t elem) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (elem << 2));
  };
  auto st4 = [&](float* base, uint32_t elem, f32x4 v) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + (elem << 2)) = v;
  };
  auto lds_off = [&](const void* q) __attribute__((always_inline)) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q; };
  int row_m[MT], row_seq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < p.M ? m : p.M - 1;
    row_m[mt] = m;
    row_seq[mt] = m / p.rows_per_seq;
  }
  // Fragment of k-chunk c (32 k-values) of rows mt*16 + l15: the 16-byte piece (c*4 + g) ^ l15 of the row (XOR swizzle).  c*4 has
  // no bits below 2 and g, l15 none above 3, so (c*4 + g) ^ l15 = (c >> 2) * 16 + (((c & 3) * 4) ^ (g ^ l15)): FOUR lane-dependent
  // offsets (c & 3) plus immediates, instead of one address register per k-chunk (16 of them, live across the whole kernel).
  uint32_t aswz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aswz[i] = (uint32_t)(((i * 4) ^ (g ^ l15)) << 4);
  // element offset inside a panel row of this lane's 4 output columns of tile 0 (XOR swizzle of the 16-byte piece, as above)
  const int pswz = (((W4 * 4 + g) ^ l15) << 3) | (J0 * 4);

  // The GEMM loops are ROLLED (bodies of PF stages with compile-time ring slots and swizzle phases, a running row pointer):
  // fully unrolled, a 32-stage group GEMM is one 320-MFMA block in which hipcc stops tying accumulator inputs to outputs and
  // spills the accumulators themselves.
  //
  // acc[mt] += P[:, 0 : 64*NKS] x (the next NKS stream stages)^T for ONE 128-column tile; NKS % PF == 0.
  // Half-stage software pipeline: while the MT MFMAs of k-chunk c issue, the MT fragments of chunk c+1 are read (the read behind
  // the last chunk runs past the tile's K range into the next panel row: valid LDS, never used).
  // `fill_c`: the caller has independent vector arithmetic (the GELU of the previous hidden chunk) in the same scheduling region;
  // two of its instructions are slotted behind every MFMA (an MFMA holds the matrix pipe for 16 cycles and its issue slot for 4).
  auto gemm_tile = [&](f32x4(&acc)[MT], const h16_t* P, int pld, auto nks_c, bool swap, auto fill_c) __attribute__((always_inline)) {
    constexpr int NKS = decltype(nks_c)::value;
    constexpr int FILL = decltype(fill_c)::value;
    static_assert(NKS % PF == 0 && PF == 4, "ring phase");
    const char* rp = reinterpret_cast<const char*>(P) + l15 * pld * 2;
    const int rstep = 32 * pld;   // bytes between the 16-row blocks of a panel
    h16x8 a[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[0] + mt * rstep);
#pragma unroll
    for (int it = 0; it < NKS / PF; ++it) {
#pragma unroll
      for (int u = 0; u < 2 * PF; ++u) {   // half stage u of this body: k-chunk 8*it + u, ring slot u >> 1
        const int slot = u >> 1, kk = u & 1;
        if (kk == 0) w_wait(slot);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[(u + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[(u + 1) & 3] + ((u + 1) >> 2) * 256 + mt * rstep);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (swap) acc[mt] = A2P_MFMA16(a[u & 1][mt], wr[slot][kk], acc[mt]);
          else acc[mt] = A2P_MFMA16(wr[slot][kk], a[u & 1][mt], acc[mt]);
        }
        if (kk == 1) w_issue(slot);
        // issue order of the half stage: MFMA, fragment read, MFMA, fragment read, ...
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if constexpr (FILL > 0) __builtin_amdgcn_sched_group_barrier(0x402, FILL, 0);   // VALU + transcendental
        }
        if (kk == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // ... then the stage's two weight loads
      }
      rp += 512;   // 8 k-chunks = two 256-byte swizzle periods
      __builtin_amdgcn_sched_barrier(0);   // bodies are scheduled one by one (rolled, hipcc drains vmcnt at every loop header)
    }
  };
  // acc[t] += P[:, 0 : 64*NKS] x stages^T for all NT tiles, k-major (stage = ks*NT + t): the A fragments of a k-step are read
  // once for the NT tiles, one k-step ahead, spread over the stages of the current k-step.  Same per-tile k-order as the
  // tile-major form: same bits.  Body = two k-steps (2*NT stages, a multiple of PF).
  auto gemm_group = [&](f32x4(&acc)[NT][MT], const h16_t* P, int pld, auto nks_c) __attribute__((always_inline)) {
    constexpr int NKS = decltype(nks_c)::value;
    static_assert(NKS % 2 == 0 && (2 * NT) % PF == 0, "ring phase");
    constexpr int RPS = (2 * MT + NT - 1) / NT;   // fragment reads per stage
    const char* rp = reinterpret_cast<const char*>(P) + l15 * pld * 2;
    const int rstep = 32 * pld;
    h16x8 a[2][2][MT];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[0][kk][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[kk] + mt * rstep);
#pragma unroll
    for (int it = 0; it < NKS / 2; ++it) {
#pragma unroll
      for (int j = 0; j < 2; ++j)      // k-step 2*it + j: fragments in a[j], the next k-step's go to a[j ^ 1]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int slot = (j * NT + t) % PF;
          w_wait(slot);
          int nread = 0;
#pragma unroll
          for (int r = t * RPS; r < (t + 1) * RPS && r < 2 * MT; ++r) {
            const int cn = 2 * (j + 1) + r / MT;   // k-chunk of the next k-step, relative to this body's first chunk
            a[j ^ 1][r / MT][r % MT] = *reinterpret_cast<const h16x8*>(rp + aswz[cn & 3] + (cn >> 2) * 256 + (r % MT) * rstep);
            ++nread;
          }
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[t][mt] = A2P_MFMA16(wr[slot][kk], a[j][kk][mt], acc[t][mt]);
          w_issue(slot);
#pragma unroll
          for (int i = 0; i < 2 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < nread) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
      rp += 256;   // two k-steps = 4 k-chunks = one swizzle period
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto init_bias_o = [&](f32x4(&acc)[MT], const float* bias_lds, int t) __attribute__((always_inline)) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + obase(t) + g * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = b;
  };
  auto init_bias_ot = [&](f32x4(&acc)[MT], const float* bias_lds, int t) __attribute__((always_inline)) {
    const float b = bias_lds[obase(t) + l15];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{b, b, b, b};
  };
  auto init_bias = [&](f32x4(&acc)[MT], const float* bias_lds) __attribute__((always_inline)) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + W4 * 32 + g * 8 + J0 * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = b;
  };

  // ---- kernel start: attention-output panel + aux block by LDS-DMA, weight ring primed ----------------------------------
  if constexpr (MODE != CHAIN_PRE) {
    constexpr int RPI = 64 / CPR;
    for (int r0 = wid * RPI; r0 < BM; r0 += NW * RPI) {
      const int row = r0 + lane / CPR, pos = lane % CPR;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      m = (p.src_rows > 0 && m >= p.src_rows) ? m - p.src_rows : m;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ain + (int64_t)m * p.ld_ain + ((pos ^ (row & 15)) << 3)),
                                       (__attribute__((address_space(3))) void*)(panelA + r0 * D), 16, 0, 0);
    }
  }
  for (int kb = wid; kb < p.aux_kb; kb += NW) chain_glds16(p.aux + kb * 256 + lane * 4, aux + kb * 256);
#pragma unroll
  for (int i = 0; i < PF; ++i) w_issue(i);
  // the DMA pieces have landed for this wave (the compiler is free to order the ring's first loads in front of them, so the
  // wait is for everything: once per kernel) ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  chain_bar();   // ... and for every other wave
  stamp(1);

  // ---- epilogues --------------------------------------------------------------------------------------------------------
  // FiLM affine + residual, IN PLACE: R[t][mt] = x_old + (scale + 1) * (R + bias) + shift   (transformer_modules.py:122-124,193)
  // x_old comes from `xs` in the given layout; reading it here instead of at kernel start keeps 16*MT registers free during the GEMM
  auto film_res = [&](f32x4(&R)[NT][MT], const float* bias, const float* film, const float* xs, int tiled, bool use_src) __attribute__((always_inline)) {
    // fence: the operand loads below do not depend on the GEMM, and left alone hipcc issues them INSIDE the GEMM that precedes them
    // (up to 3*MT*NT*4 registers on top of the accumulators: 0.7-1.2 KB of scratch per lane at 80 rows)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t xb[MT], fb[MT];
    const uint32_t ts = x_tstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ms = (use_src && p.src_rows > 0 && row_m[mt] >= p.src_rows) ? row_m[mt] - p.src_rows : row_m[mt];
      xb[mt] = x_rbase(ms, tiled);
      fb[mt] = (uint32_t)row_seq[mt] * (uint32_t)p.film_seq_stride + (uint32_t)col_of(0);
    }
    // operands in batches of at most 3 rows of a tile (9 f32x4 = 36 registers in flight next to the 16*MT accumulators and the
    // weight ring; whole-tile batches spilled at 80 rows), all loads of a batch issued before its arithmetic
    constexpr int RB = MT <= 3 ? MT : (MT + 1) / 2;
    // (`film` is never NULL here: the host routes FiLM-less chains to kernels_chain.h -- a uniform branch around this region
    // made hipcc spill ~50 accumulator registers across it)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + col_of(t));
#pragma unroll
      for (int m1 = 0; m1 < MT; m1 += RB) {
        f32x4 xo[RB], sc[RB], sh[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int mt = m1 + i < MT ? m1 + i : MT - 1;
          xo[i] = ld4(xs, xb[mt] + t * ts);
          sc[i] = ld4(film, fb[mt] + t * 128);
          sh[i] = ld4(film, fb[mt] + t * 128 + (uint32_t)p.film_shift_off);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int mt = m1 + i;
          if (mt >= MT) break;
          const f32x4 y = R[t][mt] + b, s1 = sc[i] + 1.0f;
          f32x4 xr = xo[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] += fmaf(s1[e], y[e], sh[i][e]);
          R[t][mt] = xr;
          // the RESULT is pinned here: hipcc otherwise sinks this arithmetic to the first use of the rows (the LayerNorm sums) and
          // keeps every loaded operand of every batch alive until then (128 registers at 80 rows)
          asm volatile("" : "+v"(R[t][mt]));
        }
        __builtin_amdgcn_sched_barrier(0);   // one batch at a time
      }
    }
  };
  float ln_mean[MT], ln_rstd[MT];
  auto group_partials = [&](const float* q) __attribute__((always_inline)) {
    return ((q[0] + q[BM]) + (q[2 * BM] + q[3 * BM])) + ((q[4 * BM] + q[5 * BM]) + (q[6 * BM] + q[7 * BM]));
  };
  auto ln_stats = [&](const f32x4(&R)[NT][MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) v += (R[t][mt][0] + R[t][mt][1]) + (R[t][mt][2] + R[t][mt][3]);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) red[wid * BM + mt * 16 + l15] = v;
    }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = mt * 16 + l15;
      ln_mean[mt] = group_partials(red + r) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = R[t][mt][e] - ln_mean[mt];
          q = fmaf(dlt, dlt, q);
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (g == 0) red[8 * BM + wid * BM + r] = q;
    }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float var = group_partials(red + 8 * BM + mt * 16 + l15) * (1.0f / D);
      ln_rstd[mt] = 1.0f / sqrtf(var + 1e-5f);
    }
  };
  // normalised (optionally rotated) rows -> 16-bit A panel
  auto ln_write = [&](const f32x4(&R)[NT][MT], const float* gamma, const float* beta, auto rope_c) __attribute__((always_inline)) {
    constexpr bool ROPE = decltype(rope_c)::value;
    // gamma / beta of all tiles in one batch; the rotary entries (MT per tile) one tile ahead of their use
    f32x4 ga[NT], be[NT], cs[2][ROPE ? MT : 1];
    int pos[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pos[mt] = row_m[mt] - row_seq[mt] * p.rows_per_seq;
    auto load_cs = [&](int t) __attribute__((always_inline)) {
      if constexpr (ROPE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          cs[t & 1][mt] = ld4(reinterpret_cast<const float*>(p.cst), ((uint32_t)(col_of(t) >> 2) * (uint32_t)p.cs_npos + (uint32_t)pos[mt]) << 2);
      }
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ga[t] = *reinterpret_cast<const f32x4*>(gamma + col_of(t));
      be[t] = *reinterpret_cast<const f32x4*>(beta + col_of(t));
    }
    load_cs(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_barrier(0);     // at most two tiles' rotary entries in flight
      if (t + 1 < NT) load_cs(t + 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float rs = ln_rstd[mt], nm = -ln_mean[mt] * rs;
        float v0 = fmaf(fmaf(R[t][mt][0], rs, nm), ga[t][0], be[t][0]);
        float v1 = fmaf(fmaf(R[t][mt][1], rs, nm), ga[t][1], be[t][1]);
        float v2 = fmaf(fmaf(R[t][mt][2], rs, nm), ga[t][2], be[t][2]);
        float v3 = fmaf(fmaf(R[t][mt][3], rs, nm), ga[t][3], be[t][3]);
        if constexpr (ROPE) {
          const f32x4 c = cs[t & 1][mt];
          const float r0 = fmaf(v0, c[0], -(v1 * c[1])), r1 = fmaf(v1, c[0], v0 * c[1]);
          const float r2 = fmaf(v2, c[2], -(v3 * c[3])), r3 = fmaf(v3, c[2], v2 * c[3]);
          v0 = r0; v1 = r1; v2 = r2; v3 = r3;
        }
        *reinterpret_cast<h16x4*>(panelA + (mt * 16 + l15) * D + t * 128 + pswz) = h16x4{(h16_t)v0, (h16_t)v1, (h16_t)v2, (h16_t)v3};
      }
    }
    chain_bar();   // the panel is complete before any wave's fragment reads
  };
  auto load_x = [&](f32x4(&R)[NT][MT]) __attribute__((always_inline)) {
    const float* xs = p.xsrc ? p.xsrc : p.x;
    const uint32_t ts = x_tstride(p.x_in_tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ms = (p.src_rows > 0 && row_m[mt] >= p.src_rows) ? row_m[mt] - p.src_rows : row_m[mt];
      const uint32_t xb = x_rbase(ms, p.x_in_tiled);
#pragma unroll
      for (int t = 0; t < NT; ++t) R[t][mt] = ld4(xs, xb + t * ts);
    }
  };
  auto store_x = [&](const f32x4(&R)[NT][MT], int tiled) __attribute__((always_inline)) {
    const uint32_t ts = x_tstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (m0 + mt * 16 + l15 >= p.M) continue;
      const uint32_t xb = x_rbase(row_m[mt], tiled);
#pragma unroll
      for (int t = 0; t < NT; ++t) st4(p.x, xb + t * ts, R[t][mt]);
    }
  };
  // D-deep GEMM over `ntiles` output tiles with a per-tile 16-bit store (kernels_chain.h gemm_store, 8-wave forms): row-major
  // tiles leave in pairs as 16 rows x 64 contiguous bytes per instruction, V^T tiles as 16-byte pieces of 8 consecutive frames
  // The tile loop is unrolled (compile-time tile count): across a loop back-edge hipcc waits for ALL outstanding ring loads
  // (s_waitcnt vmcnt(0) at every loop header), i.e. one exposed L2 round trip per tile.
  auto gemm_store = [&](auto ntiles_c, const float* bias_lds, h16_t* out, int64_t ldo, bool transposed) __attribute__((always_inline)) {
    constexpr int ntiles = decltype(ntiles_c)::value;
    // V^T tiles are transposed through a wave-private slice of the idle hidden-chunk buffer and leave as 16-byte pieces (8
    // consecutive frames of one column).  Frame counts that are not a multiple of 8 take the generation-1 kernels (host).
    constexpr int VP = (CW * BM / 8 + 63) / 64;
    h16_t* const stg = panelH + wid * (CW * BM);
    uint32_t voff[VP];
    bool vok[VP];
    if (transposed) {
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int q = lane + 64 * i, c = q / (BM / 8), m = m0 + (q % (BM / 8)) * 8;
        const int sq = m / p.rows_per_seq;
        vok[i] = q < CW * BM / 8 && m < p.M;
        voff[i] = (uint32_t)sq * (uint32_t)p.vt_seq_stride + (uint32_t)(m - sq * p.rows_per_seq) + (uint32_t)c * (uint32_t)ldo;
      }
    }
    h16x4 held[MT];
#pragma unroll
    for (int t = 0; t < ntiles; ++t) {
      f32x4 acc[MT];
      if (!transposed) init_bias_o(acc, bias_lds, t);
      else init_bias_ot(acc, bias_lds, t);
      C2_FENCE();
      gemm_tile(acc, panelA, D, std::integral_constant<int, KS>{}, transposed, std::integral_constant<int, 0>{});
      C2_FENCE();
      if (!transposed) {
        if ((t & 1) == 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[mt];
            held[mt] = h16x4{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[mt];
            const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            const int hsw = (l15 >> 2) & 1;   // half-row swizzle of the [16][32] staging tile (scratch/lds_probe: no 4-way conflict)
            asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" ::"v"(lds_off(stg + l15 * 32 + hsw * 16 + g * 4)),
                         "v"(lds_off(stg + l15 * 32 + (hsw ^ 1) * 16 + g * 4)), "v"(held[mt]), "v"(o)
                         : "memory");
            h16x8 w;
            const int prow = lane >> 2, pp = lane & 3;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(w)
                         : "v"(lds_off(stg + prow * 32 + (((pp >> 1) ^ ((prow >> 2) & 1)) * 2 + (pp & 1)) * 8))
                         : "memory");
            const int m = m0 + mt * 16 + (lane >> 2);
            if (m < p.M)
              *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + (((uint32_t)m * (uint32_t)ldo + (uint32_t)(obase(t - 1) + (lane & 3) * 8)) << 1)) = w;
          }
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[mt];
          const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
          asm volatile("ds_write_b64 %0, %1" ::"v"(lds_off(stg + l15 * BM + mt * 16 + g * 4)), "v"(o) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < VP; ++i) {
          h16x8 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(stg + (lane + 64 * i) * 8)) : "memory");
          if (vok[i]) *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + ((voff[i] + (uint32_t)obase(t) * (uint32_t)ldo) << 1)) = v;
        }
      }
    }
  };
  // norm1 -> rotary -> [Q|K] ; norm1 -> V^T     (aux: bias_qk at aq, bias_v right behind).  `R` holds the finished residual rows;
  // they are stored (x_out layout) between the two LayerNorm writes, as soon as nothing needs them in registers any more.
  auto pre_work = [&](f32x4(&R)[NT][MT], const float* aq, bool write_x) __attribute__((always_inline)) {
    C2_FENCE();
    ln_stats(R);
    C2_FENCE();
    ln_write(R, p.lnB_g, p.lnB_b, std::true_type{});
    C2_FENCE();
    if (write_x) store_x(R, p.x_out_tiled);
    C2_FENCE();
    stamp(9);
    // the 16*MT row registers are free during the [Q|K] GEMM (80 rows: acc + fragments + ring + the rows did not fit 256)
    gemm_store(std::integral_constant<int, 2 * NT>{}, aq, p.qk_out, p.ld_qk, false);
    chain_bar();   // every wave is done reading the rotated panel
    C2_FENCE();
    stamp(10);
    if (write_x) {   // back from where store_x left them (same lanes, program order)
      const uint32_t ts = x_tstride(p.x_out_tiled);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint32_t xb = x_rbase(row_m[mt], p.x_out_tiled);
#pragma unroll
        for (int t = 0; t < NT; ++t) R[t][mt] = ld4(p.x, xb + t * ts);
      }
    } else {
      load_x(R);     // MODE_PRE: the rows were never modified
    }
    C2_FENCE();
    ln_write(R, p.lnB_g, p.lnB_b, std::false_type{});
    C2_FENCE();
    stamp(11);
    gemm_store(std::integral_constant<int, NT>{}, aq + 2 * D, p.vt_out, p.ld_vt, true);
    stamp(12);
  };

  // =========================================================================================================================
  f32x4 R[NT][MT];
  if constexpr (MODE == CHAIN_PRE) {
    load_x(R);
    pre_work(R, aux, false);
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) R[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    C2_FENCE();
    if constexpr (!(C2_ABL & 1)) gemm_group(R, panelA, D, std::integral_constant<int, KS>{});                      // out_proj of the attention that produced `ain`
    C2_FENCE();
    stamp(2);
    if constexpr (!(C2_ABL & 2)) film_res(R, p.bias_o, p.film_o, p.xsrc ? p.xsrc : p.x, p.x_in_tiled, true);
    C2_FENCE();
    stamp(3);
    if constexpr (!(C2_ABL & 4)) ln_stats(R);                                                                       // (its barriers also order the panel rewrite behind every wave's out_proj reads)
    C2_FENCE();
    if constexpr (MODE == CHAIN_MID) {
      if constexpr (!(C2_ABL & 4)) ln_write(R, p.lnA_g, p.lnA_b, std::true_type{});
      C2_FENCE();
      store_x(R, p.x_out_tiled);
      C2_FENCE();
      if constexpr (!(C2_ABL & 8)) gemm_store(std::integral_constant<int, NT>{}, aux, p.q_out, p.ld_q, false);
    } else {
      ln_write(R, p.lnA_g, p.lnA_b, std::false_type{});
      C2_FENCE();
      store_x(R, p.x_out_tiled);            // parked: the feed-forward block runs without the residual rows in registers
      C2_FENCE();
      stamp(4);
      // Feed forward, split-K over the 8 hidden chunks: linear1 chunk -> GELU -> LDS -> linear2 partial.
      // (Measured and dropped: linear1 of chunk h+1 issued in the same scheduling region as the GELU of chunk h with a
      // double-buffered hidden chunk -- one barrier per chunk instead of two, vector work slotted behind the MFMAs: bit-identical,
      // +5 % kernel time at 48 rows and register spills at 80, profiles/r03_chain2_bench.txt.)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) R[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < FT; ++h) {
        f32x4 acc[MT];
        init_bias(acc, aux + h * 128);
        C2_FENCE();
        gemm_tile(acc, panelA, D, std::integral_constant<int, KS>{}, false, std::integral_constant<int, 0>{});
        C2_FENCE();
        if (h > 0) chain_bar();   // every wave finished the linear2 partial of the previous chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[mt];
          *reinterpret_cast<h16x4*>(panelH + (mt * 16 + l15) * HLD + pswz) =
              h16x4{(h16_t)act_gelu_fast(v[0]), (h16_t)act_gelu_fast(v[1]), (h16_t)act_gelu_fast(v[2]), (h16_t)act_gelu_fast(v[3])};
        }
        chain_bar();              // the hidden chunk is complete
        C2_FENCE();
        gemm_group(R, panelH, HLD, std::integral_constant<int, 2>{});
        C2_FENCE();
      }
      // the parked rows come back from where store_x left them (same workgroup, same lanes: program order makes them visible)
      stamp(5);
      film_res(R, p.bias_2, p.film_f, p.x, p.x_out_tiled, false);
      C2_FENCE();
      stamp(6);
      if (p.has_next) {
        pre_work(R, aux + FT * 128, true);
      } else {
        store_x(R, p.x_out_tiled);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's run-ahead loads target this wave's registers
}

// Stream leader (ChainP::n_pf): walks the weight stream with up to 63 KiB in flight per participating wave.  Every load targets
// the same scratch register (loads return in order; nothing is ever consumed), the hardware's 6-bit vmcnt is the throttle.
__device__ __forceinline__ void chain2_stream_leader(const ChainP& p) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wid >= p.pf_waves) return;
  const uint32_t total = (uint32_t)p.n_stages * 16384u, step = (uint32_t)p.pf_waves * 1024u;
  f32x4 sink;
  for (uint32_t off = (uint32_t)wid * 1024u + (uint32_t)lane * 16u; off < total; off += step)
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink) : "v"(off), "s"(p.stream));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D, int MT, int MODE>
__global__ __launch_bounds__(512, 1) void chain2_kernel(const ChainP p) {
  __shared__ __attribute__((aligned(16))) h16_t smem[Chain2Lds<D, MT>::ELEMS];
  if ((int)blockIdx.x < p.n_pf) {   // workgroup-uniform
    chain2_stream_leader(p);
    return;
  }
  chain2_body<D, MT, MODE>(p, smem, ((int)blockIdx.x - p.n_pf) * (16 * MT));
}
#pragma clang fp contract(fast)
