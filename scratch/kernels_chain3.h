// EXPERIMENT (round 4; scratch, not product -- built only by scratch/chain3_bench.hip): row-panel chain kernels, third generation:
// TWO workgroups per CU.  Bit-identical to kernels_chain.h (14 GPU test cases at 0.0 while it was wired into the library, commit
// "Chain kernels generation 3"), measured EQUAL-TO-SLOWER inside the step (B=32: 207-225 vs 200 us per chain launch; B=16 equal;
// B=8 and the body model slower) and removed from the library again -- profiles/r04_chain3_bench*.txt, docs/lab_notebook_r1_r4.md section 4.1c.
//
// The same three chains as kernels_chain.h (PRE / MID / POST of FiLMTransformerDecoderLayer.forward,
// transformer_modules.py:178-267) and the same bits.  What rounds 1-3 measured: a chain launch spends
// ~45 % of its time in row-local epilogues (FiLM, LayerNorm, GELU, rotary, stores) with the matrix pipe idle, and ONE workgroup
// per CU (the LDS weight ring of generation 1, the 8 x 256-register waves of generation 2) keeps all its waves in the same phase:
// its barriers make sure of that.  Here a workgroup is small enough that two of them share a CU -- 4 waves of <= 256 registers,
// <= 80 KiB of LDS -- and the hardware interleaves them: while one normalises / stores on the vector ALU, the other owns the
// matrix pipe, and inside the GEMM loops the two waves of a SIMD cover each other's fragment reads, weight loads and waits.
//
//   * Weights go straight from L2 into VGPRs (generation 2's finding: a weight fragment has exactly one consumer wave, an LDS
//     ring buys nothing).  The stream is packed so that one global_load_dwordx4 of a wave is one MFMA operand; a wave owns 32 of a
//     tile's 128 output columns (two 16-column sub-tiles, ADJACENT in the output: 8 contiguous columns per lane), i.e. 4 loads per
//     16 KiB stage, and a PF-stage register ring runs ahead across tiles, GEMMs and epilogues.
//   * No LDS ring: panel (48 x 512) + hidden chunk + LayerNorm partials + bias block = 73 KiB at d = 512.
//   * The fp32 residual rows are not register-resident across the feed-forward block (96 registers at 48 rows next to the 96
//     accumulators of linear2): FiLM + residual folds them into the out_proj accumulators IN PLACE, the result is normalised into
//     the panel, stored, and read back by the second FiLM epilogue (one extra 2 KiB store + load per row, through L2).
//   * Column ownership, accumulation order per output element, the 8-partial LayerNorm tree and the epilogue arithmetic are those
//     of kernels_chain.h's 4-wave shape: bit-identical results (tests/test_hip_round3.py), so the host may pick either.
//
// Bound: every workgroup streams all weights of its chain, 16 KiB per 48 rows x 128 columns x 64 k = 192 MFMA cycles per SIMD,
// and the L2 -> CU path delivers 64 B/clk: two workgroups in their GEMM loops are bound by that path (256 cycles per stage each =
// 75 % of the matrix peak); what the second workgroup buys is everything else.
#pragma once
#include "../audio2photoreal_amd/csrc/kernels_chain.h"

// the experiment's extra launch parameters (not part of the product's ChainP)
struct Chain3P : ChainP {
  int phase_us, phase_blocks;   // start delay (us) of the workgroups placed second on their CU among the first phase_blocks of the launch
  int x_prefetch;               // touch the residual rows at kernel start
};

#pragma clang fp contract(off)

#ifndef CHAIN3_PF
#define CHAIN3_PF 4   // weight stages in flight per wave (register ring: 16 VGPRs per stage); every GEMM consumes a multiple of it
#endif
// phase boundary: nothing is scheduled across (each phase of a chain is its own scheduling region -- in one region hipcc
// interleaves the address arithmetic and loads of later phases with the GEMM in front of them and runs out of registers)
#define C3_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int D, int MT>
struct Chain3Lds {
  static constexpr int BM = 16 * MT, AUX_F = 2560;
  static constexpr int ELEMS = BM * D + BM * 128 + 32 * BM + 2 * AUX_F;   // 16-bit elements: panelA, panelH, LN partials, aux
  static_assert(ELEMS * 2 <= 80 * 1024, "two workgroups must fit the 160 KiB of a CU");
};

// One 16 KiB stage as MFMA operands: [wave 0..3][sub-tile J 0..1][k-chunk 0..1][lane 0..63][8 k-values] -- lane (l15, g) of wave w
// gets W[col(w, J, l15)][k0 + (kk*4 + g)*8 .. +8], the B operand of v_mfma_f32_16x16x32 for k-chunk kk.  Column ownership as
// kernels_chain.h's 4-wave shape: sub-tile row i = 4*g' + e is output column row0 + w*32 + g'*8 + J*4 + e.
__global__ __launch_bounds__(256) void chain3_pack_kernel(const ChainPackDesc* __restrict__ descs, h16_t* __restrict__ dst) {
  const ChainPackDesc d = descs[blockIdx.x];
  uint4* out = reinterpret_cast<uint4*>(dst + (int64_t)blockIdx.x * CHAIN_STAGE_ELEMS);
  for (int q = threadIdx.x; q < 1024; q += 256) {
    const int w = q >> 8, J = (q >> 7) & 1, kk = (q >> 6) & 1, lane = q & 63, i = lane & 15, g = lane >> 4;
    const int row = d.row0 + w * 32 + (i >> 2) * 8 + J * 4 + (i & 3);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < d.nrows) v = *reinterpret_cast<const uint4*>(d.W + (int64_t)row * d.ldw + d.k0 + (kk * 4 + g) * 8);
    out[q] = v;
  }
}

template <int D, int MT, int MODE>
__device__ __forceinline__ void chain3_body(const Chain3P& p, h16_t* const smem, const int m0) {
  constexpr int NJ = 2, CW = 32, BM = 16 * MT, CPR = D / 8, NT = D / 128, KS = D / 64, FT = 8, HLD = 128, AUX_F = 2560, PF = CHAIN3_PF;
  static_assert(PF == 4 || PF == 2, "ring depth");
  static_assert(KS % PF == 0 && (2 * NT) % PF == 0, "every GEMM must consume a multiple of the register ring");
  h16_t* const panelA = smem;
  h16_t* const panelH = panelA + BM * D;
  float* const red = reinterpret_cast<float*>(panelH + BM * HLD);   // [2][8][BM] LayerNorm partial sums, one per (wave, sub-tile)
  float* const aux = red + 16 * BM;                                   // [AUX_F] per-tile biases
  const int tid = threadIdx.x, lane = tid & 63, W4 = tid >> 6, l15 = lane & 15, g = lane >> 4;
#if defined(C3_STAMPS) || defined(A2P_STAMPS)   // scratch/chain3_bench.hip, scratch/phase_probe.py: 100 MHz phase stamps of workgroups 0 and 301 into p.fin_out
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 301)) reinterpret_cast<unsigned long long*>(p.fin_out)[(blockIdx.x ? 32 : 0) + i] = wall_clock64();
  };
#else
  auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
  stamp(0);

  // ---- weight stream: a register ring of PF stages, 4 fragments (2 sub-tiles x 2 k-chunks) per stage --------------------
  // Ordinary (compiler-visible) loads: hipcc places the s_waitcnt vmcnt(N) itself and knows that a ring register is not valid
  // before it (as inline asm the loads of generation 2 worked until the register allocator copied a ring register across a loop
  // back-edge BEFORE the hand-written wait).
  // Buffer loads through one wave-uniform descriptor: the lane-constant part of the address is ONE VGPR, the stage advance lives
  // in an SGPR (soffset) and the four fragments of a stage are immediates -- no vector address arithmetic in the GEMM loops
  // (as flat loads hipcc kept a 64-bit address pair per stage in flight: 2 x PF x ... registers and a v_lshl_add_u64 each).
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16_t*>(p.stream), 0, 0x7fffffff, 0x00020000);
  const int wvo = (W4 * 256 + lane) * 16;   // byte offset of this lane's 16 bytes of fragment (J 0, k-chunk 0) inside a stage
  int wso = 0;                              // byte offset of the next stage to load (uniform)
  h16x8 wr[PF][NJ][2];
  auto w_issue = [&](int slot) __attribute__((always_inline)) {
    // default cache policy: every workgroup of the launch walks the same stream, L2 serves all but the first
    wr[slot][0][0] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo, wso, 0));
    wr[slot][0][1] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo + 1024, wso, 0));
    wr[slot][1][0] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo + 2048, wso, 0));
    wr[slot][1][1] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo + 3072, wso, 0));
    wso += 16384;
  };

  auto w_prime = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PF; ++i) w_issue(i);
  };

  // ---- helpers ----------------------------------------------------------------------------------------------------------
  // first of the 4 consecutive output columns this lane holds of sub-tile (tile t, half j); the two halves of a lane are adjacent
  auto col_of = [&](int t, int j) __attribute__((always_inline)) { return t * 128 + W4 * 32 + g * 8 + j * 4; };
  // Residual-row addressing with 32-bit element offsets (a forward holds < 2^32 bytes of rows): offset of this lane's 4 columns of
  // sub-tile (t, j) of row m = x_rbase(m, layout) + t * x_tstride(layout) + j * x_jstride(layout).  Tiled layout
  // (ChainP::x_in_tiled): per 16-row block the D/16 chunks (tile t, 32-column group W4, half j) of 1 KiB each, chunk =
  // [g][row & 15][4 floats]; row-major: m * D + col_of(t, j).
  auto x_rbase = [&](int m, int tiled) __attribute__((always_inline)) -> uint32_t {
    const uint32_t a = (uint32_t)(((m >> 4) * (D / 16) + W4 * 2) * 256 + (g * 16 + (m & 15)) * 4);
    const uint32_t b = (uint32_t)(m * D + W4 * 32 + g * 8);
    return tiled ? a : b;
  };
  auto x_tstride = [&](int tiled) __attribute__((always_inline)) -> uint32_t { return tiled ? 2048u : 128u; };
  auto x_jstride = [&](int tiled) __attribute__((always_inline)) -> uint32_t { return tiled ? 256u : 4u; };
  // global accesses as (uniform base pointer) + (32-bit BYTE offset): the form global_load/store take as SGPR base + VGPR offset
  auto ld4 = [&](const float* base, uint32_t elem) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (elem << 2));
  };
  auto st4 = [&](float* base, uint32_t elem, f32x4 v) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + (elem << 2)) = v;
  };
  auto lds_off = [&](const void* q) __attribute__((always_inline)) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q; };
  // The row indices are re-materialised through an opaque move at the start of every epilogue: hipcc otherwise computes the
  // (loop-invariant) 64-bit addresses of an epilogue's ~70 global accesses in front of the GEMM loops that precede it and
  // spills them across those loops
  auto opaque = [&](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
  int row_m[MT], row_seq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < p.M ? m : p.M - 1;
    row_m[mt] = m;
    row_seq[mt] = m / p.rows_per_seq;
  }
  // Fragment of k-chunk c (32 k-values) of rows mt*16 + l15: the 16-byte piece (c*4 + g) ^ l15 of the row (XOR swizzle) =
  // (c >> 2) * 16 + (((c & 3) * 4) ^ (g ^ l15)): FOUR lane-dependent offsets plus immediates.
  uint32_t aswz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aswz[i] = (uint32_t)(((i * 4) ^ (g ^ l15)) << 4);
  // element offset inside a panel row of this lane's 8 output columns of tile 0 (XOR swizzle of the 16-byte piece)
  const int pswz = ((W4 * 4 + g) ^ l15) << 3;

  // acc[j][mt] += P[:, 0 : 64*NKS] x (the next NKS stream stages)^T for ONE 128-column tile; NKS % PF == 0.
  // Half-stage software pipeline: while the 2*MT MFMAs of k-chunk c issue, the MT fragments of chunk c+1 are read (the read behind
  // the last chunk runs past the tile's K range into the next panel row: valid LDS, never used).
  // `tail_c`: the GEMM is followed by an epilogue, not by another GEMM: its last PF stages do not refill the ring, so that the
  // ring's registers are free for the epilogue's operand batches; the epilogue primes the ring again (w_prime) when it is done.
  auto gemm_tile = [&](f32x4(&acc)[NJ][MT], const h16_t* P, int pld, auto nks_c, bool swap, auto tail_c) __attribute__((always_inline)) {
    constexpr int NKS = decltype(nks_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
    static_assert(NKS % PF == 0, "ring phase");
    const char* rp = reinterpret_cast<const char*>(P) + l15 * pld * 2;
    const int rstep = 32 * pld;   // bytes between the 16-row blocks of a panel
    h16x8 a[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[0] + mt * rstep);
#pragma unroll
    for (int it = 0; it < NKS / PF; ++it) {
#pragma unroll
      for (int u = 0; u < 2 * PF; ++u) {   // half stage u of this body: k-chunk 2*PF*it + u, ring slot u >> 1
        const int slot = u >> 1, kk = u & 1;
        const int cn = (2 * PF * it + u + 1);   // next k-chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[(u + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[cn & 3] + (cn >> 2) * 256 + mt * rstep);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (swap) acc[j][mt] = A2P_MFMA16(a[u & 1][mt], wr[slot][j][kk], acc[j][mt]);
            else acc[j][mt] = A2P_MFMA16(wr[slot][j][kk], a[u & 1][mt], acc[j][mt]);
          }
        const bool refill = kk == 1 && !(TAIL && it == NKS / PF - 1);
        if (refill) w_issue(slot);
        // issue order of the half stage: MFMA, MFMA, fragment read, ...
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (refill) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // ... then the stage's four weight loads
        // every half stage is its own scheduling region: left to one region per body hipcc sinks the fragment reads of the first
        // PF stages into the half stage that consumes them (ds_read, s_waitcnt lgkmcnt(0), MFMA: un-pipelined)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // acc[t] += P[:, 0 : 64*NKS] x stages^T for all NT tiles, k-major (stage = ks*NT + t): the A fragments of a k-step are read
  // once for the NT tiles, one k-step ahead, spread over the stages of the current k-step.  Same per-tile k-order as the
  // tile-major form: same bits.  Body = two k-steps (2*NT stages, a multiple of PF).
  auto gemm_group = [&](f32x4(&acc)[NT][NJ][MT], const h16_t* P, int pld, auto nks_c, auto tail_c) __attribute__((always_inline)) {
    constexpr int NKS = decltype(nks_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;
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
      for (int j2 = 0; j2 < 2; ++j2)      // k-step 2*it + j2: fragments in a[j2], the next k-step's go to a[j2 ^ 1]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int slot = (j2 * NT + t) % PF;
          int nread = 0;
#pragma unroll
          for (int r = t * RPS; r < (t + 1) * RPS && r < 2 * MT; ++r) {
            const int cn = 4 * it + 2 * (j2 + 1) + r / MT;   // k-chunk of the next k-step
            a[j2 ^ 1][r / MT][r % MT] = *reinterpret_cast<const h16x8*>(rp + aswz[cn & 3] + (cn >> 2) * 256 + (r % MT) * rstep);
            ++nread;
          }
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int j = 0; j < NJ; ++j) acc[t][j][mt] = A2P_MFMA16(wr[slot][j][kk], a[j2][kk][mt], acc[t][j][mt]);
          const bool refill = !(TAIL && (2 * it + j2) * NT + t >= NKS * NT - PF);
          if (refill) w_issue(slot);
#pragma unroll
          for (int i = 0; i < 2 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
            if (i < nread) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (refill) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  // accumulators start at the per-column bias held in the LDS aux block (row-major consumers: 8 contiguous columns per lane)
  auto init_bias = [&](f32x4(&acc)[NJ][MT], const float* bias_lds) __attribute__((always_inline)) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_lds + W4 * 32 + g * 8);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_lds + W4 * 32 + g * 8 + 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[0][mt] = b0; acc[1][mt] = b1; }
  };
  // swapped (D = C) orientation: one bias value per lane column n = W4*32 + (l15 >> 2)*8 + j*4 + (l15 & 3)
  auto init_bias_t = [&](f32x4(&acc)[NJ][MT], const float* bias_lds) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float b = bias_lds[W4 * 32 + (l15 >> 2) * 8 + j * 4 + (l15 & 3)];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[j][mt] = f32x4{b, b, b, b};
    }
  };

  // ---- phase shift ---------------------------------------------------------------------------------------------------------
  // The two workgroups of a CU that start together (the launch's first round) would run in lock step -- both in their GEMM loops,
  // bound by the 64 B/clk weight path, then both in their epilogues with the matrix pipe idle.  The one that sits in the UPPER
  // half of the CU's LDS (HW_REG_LDS_ALLOC.lds_base != 0: it was placed second) starts p.phase_us later, so that its epilogues
  // fall under its neighbour's GEMM loops.  Later rounds start whenever a slot frees up and need no help.
  if (p.phase_us > 0 && (int)blockIdx.x < p.phase_blocks) {
    const unsigned lds_alloc = __builtin_amdgcn_s_getreg((6 << 0) | (0 << 6) | ((8 - 1) << 11));   // hwreg(HW_REG_LDS_ALLOC, 0, 8): lds_base
    if (lds_alloc != 0) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)p.phase_us * 100ull) __builtin_amdgcn_s_sleep(32);
    }
  }
  // ---- kernel start: attention-output panel + aux block by LDS-DMA, weight ring primed ----------------------------------
  if constexpr (MODE != CHAIN_PRE) {
    constexpr int RPI = 64 / CPR;
    for (int r0 = W4 * RPI; r0 < BM; r0 += 4 * RPI) {
      const int row = r0 + lane / CPR, pos = lane % CPR;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      m = (p.src_rows > 0 && m >= p.src_rows) ? m - p.src_rows : m;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ain + (int64_t)m * p.ld_ain + ((pos ^ (row & 15)) << 3)),
                                       (__attribute__((address_space(3))) void*)(panelA + r0 * D), 16, 0, 0);
    }
  }
  for (int kb = W4; kb < p.aux_kb; kb += 4) chain_glds16(p.aux + kb * 256 + lane * 4, aux + kb * 256);
  if constexpr (MODE != CHAIN_PRE) w_prime();   // (MODE_PRE: pre_work primes the ring in front of its first GEMM)
  if constexpr (MODE != CHAIN_PRE) {
    // Touch the residual rows the out_proj epilogue will read (film_res: 8 dependent batches of loads) so that they come from
    // L2 by then instead of HBM.  Every load targets one scratch register that is never read (loads return in order; the
    // compiler's own counted waits only ever over-wait because of them).
    if (p.x_prefetch) {
      const float* xs = p.xsrc ? p.xsrc : p.x;
      const uint32_t ts = x_tstride(p.x_in_tiled), js = x_jstride(p.x_in_tiled);
      f32x4 sink;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ms = (p.src_rows > 0 && row_m[mt] >= p.src_rows) ? row_m[mt] - p.src_rows : row_m[mt];
        const uint32_t xb = x_rbase(ms, p.x_in_tiled);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(sink) : "v"((xb + t * ts + j * js) << 2), "s"(xs));
      }
    }
  }
  // the DMA pieces have landed for this wave (the compiler is free to order the ring's first loads in front of them, so the
  // wait is for everything: once per kernel) ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  chain_bar();   // ... and for every other wave
  stamp(1);

  // ---- epilogues --------------------------------------------------------------------------------------------------------
  // FiLM affine + residual, IN PLACE: R[t][j][mt] = x_old + (scale + 1) * (R + bias) + shift   (transformer_modules.py:122-124,193)
  // x_old comes from `xs` in the given layout; reading it here instead of at kernel start keeps 32*MT registers free during the GEMM
  auto film_res = [&](f32x4(&R)[NT][NJ][MT], const float* bias, const float* film, const float* xs, int tiled, bool use_src) __attribute__((always_inline)) {
    // fence: the operand loads below do not depend on the GEMM, and left alone hipcc issues them INSIDE the GEMM that precedes them
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t xb[MT], fb[MT];
    const uint32_t ts = x_tstride(tiled), js = x_jstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int rm = opaque(row_m[mt]), rs = opaque(row_seq[mt]);
      const int ms = (use_src && p.src_rows > 0 && rm >= p.src_rows) ? rm - p.src_rows : rm;
      xb[mt] = x_rbase(ms, tiled);
      fb[mt] = (uint32_t)rs * (uint32_t)p.film_seq_stride + (uint32_t)col_of(0, 0);
    }
    // operands in batches of one tile (both sub-tiles: 6 f32x4 per row = 72 registers in flight at 48 rows -- the weight ring's
    // registers are free here), all loads of a batch issued before its arithmetic: NT dependent round trips per epilogue
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 b[NJ], xo[NJ][MT], sc[NJ][MT], sh[NJ][MT];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        b[j] = *reinterpret_cast<const f32x4*>(bias + col_of(t, j));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          xo[j][mt] = ld4(xs, xb[mt] + t * ts + j * js);
          sc[j][mt] = ld4(film, fb[mt] + t * 128 + j * 4);
          sh[j][mt] = ld4(film, fb[mt] + t * 128 + j * 4 + (uint32_t)p.film_shift_off);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 y = R[t][j][mt] + b[j], s1 = sc[j][mt] + 1.0f;
          f32x4 xr = xo[j][mt];
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] += fmaf(s1[e], y[e], sh[j][mt][e]);
          R[t][j][mt] = xr;
          // the RESULT is pinned here: hipcc otherwise sinks this arithmetic to the first use of the rows (the LayerNorm sums) and
          // keeps every loaded operand of every batch alive until then
          asm volatile("" : "+v"(R[t][j][mt]));
        }
      __builtin_amdgcn_sched_barrier(0);   // one batch at a time
    }
  };
  float ln_mean[MT], ln_rstd[MT];
  auto group_partials = [&](const float* q) __attribute__((always_inline)) {
    return ((q[0] + q[BM]) + (q[2 * BM] + q[3 * BM])) + ((q[4 * BM] + q[5 * BM]) + (q[6 * BM] + q[7 * BM]));
  };
  // two-pass fp32 statistics over EIGHT partials per row (one per (wave, sub-tile): kernels_chain.h's tree)
  auto ln_stats = [&](const f32x4(&R)[NT][NJ][MT]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) v += (R[t][j][mt][0] + R[t][j][mt][1]) + (R[t][j][mt][2] + R[t][j][mt][3]);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) red[(W4 * NJ + j) * BM + mt * 16 + l15] = v;
      }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = mt * 16 + l15;
      ln_mean[mt] = group_partials(red + r) * (1.0f / D);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dlt = R[t][j][mt][e] - ln_mean[mt];
            q = fmaf(dlt, dlt, q);
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (g == 0) red[8 * BM + (W4 * NJ + j) * BM + r] = q;
      }
    }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float var = group_partials(red + 8 * BM + mt * 16 + l15) * (1.0f / D);
      ln_rstd[mt] = 1.0f / sqrtf(var + 1e-5f);
    }
  };
  // normalised (optionally rotated, rotary_embedding_torch.py:46-66) rows -> 16-bit A panel, one 16-byte LDS write per (row, tile)
  auto ln_write = [&](const f32x4(&R)[NT][NJ][MT], const float* gamma, const float* beta, auto rope_c) __attribute__((always_inline)) {
    constexpr bool ROPE = decltype(rope_c)::value;
    // gamma / beta of all tiles in one batch; the rotary entries (2*MT per tile) one tile ahead of their use
    f32x4 ga[NT][NJ], be[NT][NJ], cs[2][NJ][ROPE ? MT : 1];
    int pos[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pos[mt] = opaque(row_m[mt]) - row_seq[mt] * p.rows_per_seq;
    auto load_cs = [&](int t) __attribute__((always_inline)) {
      if constexpr (ROPE) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            cs[t & 1][j][mt] = ld4(reinterpret_cast<const float*>(p.cst), ((uint32_t)(col_of(t, j) >> 2) * (uint32_t)p.cs_npos + (uint32_t)pos[mt]) << 2);
      }
    };
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ga[t][j] = *reinterpret_cast<const f32x4*>(gamma + col_of(t, j));
        be[t][j] = *reinterpret_cast<const f32x4*>(beta + col_of(t, j));
      }
    load_cs(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_barrier(0);     // at most two tiles' rotary entries in flight
      if (t + 1 < NT) load_cs(t + 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float rs = ln_rstd[mt], nm = -ln_mean[mt] * rs;
        h16x4 o[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float v0 = fmaf(fmaf(R[t][j][mt][0], rs, nm), ga[t][j][0], be[t][j][0]);
          float v1 = fmaf(fmaf(R[t][j][mt][1], rs, nm), ga[t][j][1], be[t][j][1]);
          float v2 = fmaf(fmaf(R[t][j][mt][2], rs, nm), ga[t][j][2], be[t][j][2]);
          float v3 = fmaf(fmaf(R[t][j][mt][3], rs, nm), ga[t][j][3], be[t][j][3]);
          if constexpr (ROPE) {
            const f32x4 c = cs[t & 1][j][mt];
            const float r0 = fmaf(v0, c[0], -(v1 * c[1])), r1 = fmaf(v1, c[0], v0 * c[1]);
            const float r2 = fmaf(v2, c[2], -(v3 * c[3])), r3 = fmaf(v3, c[2], v2 * c[3]);
            v0 = r0; v1 = r1; v2 = r2; v3 = r3;
          }
          o[j] = h16x4{(h16_t)v0, (h16_t)v1, (h16_t)v2, (h16_t)v3};
        }
        *reinterpret_cast<h16x8*>(panelA + (mt * 16 + l15) * D + t * 128 + pswz) = h16x8{o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
      }
    }
    chain_bar();   // the panel is complete before any wave's fragment reads
  };
  auto load_x = [&](f32x4(&R)[NT][NJ][MT], const float* xs, int tiled, bool use_src) __attribute__((always_inline)) {
    const uint32_t ts = x_tstride(tiled), js = x_jstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int rm = opaque(row_m[mt]);
      const int ms = (use_src && p.src_rows > 0 && rm >= p.src_rows) ? rm - p.src_rows : rm;
      const uint32_t xb = x_rbase(ms, tiled);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) R[t][j][mt] = ld4(xs, xb + t * ts + j * js);
    }
  };
  auto store_x = [&](const f32x4(&R)[NT][NJ][MT], int tiled) __attribute__((always_inline)) {
    const uint32_t ts = x_tstride(tiled), js = x_jstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (m0 + mt * 16 + l15 >= p.M) continue;
      const uint32_t xb = x_rbase(opaque(row_m[mt]), tiled);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) st4(p.x, xb + t * ts + j * js, R[t][j][mt]);
    }
  };
  // D-deep GEMM over `ntiles` output tiles with a per-tile 16-bit store: row-major tiles leave as one 16-byte store per lane and
  // row (8 contiguous columns: 64 contiguous bytes per row and instruction), V^T tiles are transposed through a wave-private slice
  // of the idle hidden-chunk buffer and leave as 16-byte pieces (8 consecutive frames of one column).  Frame counts that are not
  // a multiple of 8 take the generation-1 kernels (host).
  // The tile loop is unrolled (compile-time tile count): across a loop back-edge hipcc waits for ALL outstanding ring loads.
  auto gemm_store = [&](auto ntiles_c, const float* bias_lds, h16_t* out, int64_t ldo, bool transposed) __attribute__((always_inline)) {
    constexpr int ntiles = decltype(ntiles_c)::value;
    constexpr int VP = (CW * BM / 8 + 63) / 64;
    h16_t* const stg = panelH + W4 * (CW * BM);
    uint32_t voff[VP];
    bool vok[VP];
    const int lane_o = opaque(lane);
    if (transposed) {
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int q = lane_o + 64 * i, c = q / (BM / 8), m = m0 + (q % (BM / 8)) * 8;
        const int vcol = W4 * 32 + ((c & 15) >> 2) * 8 + (c >> 4) * 4 + (c & 3);   // output column of staging row c = j*16 + i
        const int sq = m / p.rows_per_seq;
        vok[i] = q < CW * BM / 8 && m < p.M;
        voff[i] = (uint32_t)sq * (uint32_t)p.vt_seq_stride + (uint32_t)(m - sq * p.rows_per_seq) + (uint32_t)vcol * (uint32_t)ldo;
      }
    }
    uint32_t orow[MT];   // element offset of this lane's first column of tile 0 in its output rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) orow[mt] = (uint32_t)opaque(row_m[mt]) * (uint32_t)ldo + (uint32_t)col_of(0, 0);
#pragma unroll
    for (int t = 0; t < ntiles; ++t) {
      f32x4 acc[NJ][MT];
      if (!transposed) init_bias(acc, bias_lds + t * 128);
      else init_bias_t(acc, bias_lds + t * 128);
      C3_FENCE();
      if (t == ntiles - 1) gemm_tile(acc, panelA, D, std::integral_constant<int, KS>{}, transposed, std::true_type{});
      else gemm_tile(acc, panelA, D, std::integral_constant<int, KS>{}, transposed, std::false_type{});
      C3_FENCE();
      if (!transposed) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (m0 + mt * 16 + l15 >= p.M) continue;
          const f32x4 v = acc[0][mt], u = acc[1][mt];
          *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + ((orow[mt] + (uint32_t)(t * 128)) << 1)) =
              h16x8{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3], (h16_t)u[0], (h16_t)u[1], (h16_t)u[2], (h16_t)u[3]};
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[j][mt];
            const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            asm volatile("ds_write_b64 %0, %1" ::"v"(lds_off(stg + (j * 16 + l15) * BM + mt * 16 + g * 4)), "v"(o) : "memory");
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < VP; ++i) {
          h16x8 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(stg + (lane + 64 * i) * 8)) : "memory");
          if (vok[i]) *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + ((voff[i] + (uint32_t)(t * 128) * (uint32_t)ldo) << 1)) = v;
        }
      }
    }
  };
  // norm1 -> rotary -> [Q|K] ; norm1 -> V^T     (aux: bias_qk at aq, bias_v right behind).  `R` holds the finished residual rows;
  // they are stored (x_out layout) between the two LayerNorm writes, as soon as nothing needs them in registers any more, and
  // read back for the un-rotated panel of the V projection (same lanes, program order; an L2 hit).
  auto pre_work = [&](f32x4(&R)[NT][NJ][MT], const float* aq, bool write_x) __attribute__((always_inline)) {
    C3_FENCE();
    ln_stats(R);
    C3_FENCE();
    ln_write(R, p.lnB_g, p.lnB_b, std::true_type{});
    C3_FENCE();
    // (the ring is primed BEFORE the row stores: vmcnt counts stores too, and loads issued behind 96 KiB of stores would wait for
    // their acknowledgement in front of the first GEMM stage)
    w_prime();
    C3_FENCE();
    if (write_x) store_x(R, p.x_out_tiled);
    C3_FENCE();
    stamp(9);
    gemm_store(std::integral_constant<int, 2 * NT>{}, aq, p.qk_out, p.ld_qk, false);
    chain_bar();   // every wave is done reading the rotated panel
    C3_FENCE();
    stamp(10);
    if (write_x) load_x(R, p.x, p.x_out_tiled, false);                 // back from where store_x left them
    else load_x(R, p.xsrc ? p.xsrc : p.x, p.x_in_tiled, true);         // MODE_PRE: the rows were never modified
    C3_FENCE();
    ln_write(R, p.lnB_g, p.lnB_b, std::false_type{});
    C3_FENCE();
    w_prime();
    stamp(11);
    gemm_store(std::integral_constant<int, NT>{}, aq + 2 * D, p.vt_out, p.ld_vt, true);
    stamp(12);
  };

  // =========================================================================================================================
  f32x4 R[NT][NJ][MT];
  if constexpr (MODE == CHAIN_PRE) {
    load_x(R, p.xsrc ? p.xsrc : p.x, p.x_in_tiled, true);
    pre_work(R, aux, false);
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) R[t][j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    C3_FENCE();
    gemm_group(R, panelA, D, std::integral_constant<int, KS>{}, std::true_type{});    // out_proj of the attention that produced `ain`
    C3_FENCE();
    stamp(2);
    film_res(R, p.bias_o, p.film_o, p.xsrc ? p.xsrc : p.x, p.x_in_tiled, true);
    C3_FENCE();
    stamp(3);
    ln_stats(R);                                                                       // (its barriers also order the panel rewrite behind every wave's out_proj reads)
    C3_FENCE();
    if constexpr (MODE == CHAIN_MID) {
      ln_write(R, p.lnA_g, p.lnA_b, std::true_type{});
      C3_FENCE();
      w_prime();
      C3_FENCE();
      store_x(R, p.x_out_tiled);
      C3_FENCE();
      gemm_store(std::integral_constant<int, NT>{}, aux, p.q_out, p.ld_q, false);
    } else {
      ln_write(R, p.lnA_g, p.lnA_b, std::false_type{});
      C3_FENCE();
      w_prime();
      C3_FENCE();
      store_x(R, p.x_out_tiled);            // parked: the feed-forward block runs without the residual rows in registers
      C3_FENCE();
      stamp(4);
      // Feed forward, split-K over the 8 hidden chunks: linear1 chunk -> GELU -> LDS -> linear2 partial.
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) R[t][j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto ffn_chunk = [&](int h, auto last_c) __attribute__((always_inline)) {
        f32x4 acc[NJ][MT];
        init_bias(acc, aux + h * 128);
        C3_FENCE();
        gemm_tile(acc, panelA, D, std::integral_constant<int, KS>{}, false, std::false_type{});
        C3_FENCE();
        if (h > 0) chain_bar();   // every wave finished the linear2 partial of the previous chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[0][mt], u = acc[1][mt];
          *reinterpret_cast<h16x8*>(panelH + (mt * 16 + l15) * HLD + pswz) =
              h16x8{(h16_t)act_gelu_fast(v[0]), (h16_t)act_gelu_fast(v[1]), (h16_t)act_gelu_fast(v[2]), (h16_t)act_gelu_fast(v[3]),
                    (h16_t)act_gelu_fast(u[0]), (h16_t)act_gelu_fast(u[1]), (h16_t)act_gelu_fast(u[2]), (h16_t)act_gelu_fast(u[3])};
        }
        chain_bar();              // the hidden chunk is complete
        C3_FENCE();
        gemm_group(R, panelH, HLD, std::integral_constant<int, 2>{}, last_c);
        C3_FENCE();
      };
      for (int h = 0; h < FT - 1; ++h) ffn_chunk(h, std::false_type{});
      ffn_chunk(FT - 1, std::true_type{});   // peeled: the last linear2 partial leaves the ring empty for the epilogue
      // the parked rows come back from where store_x left them (same workgroup, same lanes: program order makes them visible)
      stamp(5);
      film_res(R, p.bias_2, p.film_f, p.x, p.x_out_tiled, false);
      C3_FENCE();
      stamp(6);
      if (p.has_next) {
        pre_work(R, aux + FT * 128, true);
      } else {
        store_x(R, p.x_out_tiled);
      }
    }
  }
  stamp(13);
}

template <int D, int MT, int MODE>
__global__ __launch_bounds__(256, 2) void chain3_kernel(const Chain3P p) {
  __shared__ __attribute__((aligned(16))) h16_t smem[Chain3Lds<D, MT>::ELEMS];
  chain3_body<D, MT, MODE>(p, smem, (int)blockIdx.x * (16 * MT));
}
#pragma clang fp contract(fast)
