// Row-panel chain kernels, TALL form (round 5): the MID / POST chains of kernels_chain.h (FiLMTransformerDecoderLayer.forward,
// transformer_modules.py:178-267) for forwards of >= 16 sequences of the face model (d = 512), where a CU owes >= 75 rows and a
// 48-row panel streams every weight byte for too few of them (docs/lab_notebook_r1_r4.md section 4.1c: 16 KiB of weights per 192 MFMA cycles
// against the 64 B/clk L2 -> CU path; B=32 runs three rounds of 48 / 64-row panels).  Same arithmetic, same column ownership,
// same accumulation order and LayerNorm tree as the 8-wave form of kernels_chain.h -- the outputs are bit-identical
// (tests/test_hip_round5.py) -- restructured around four facts:
//
//   * PANELS OF 64 / 80 ROWS.  80 rows x 512 columns x 16 bit = 80 KiB of LDS: there is no room left for the 80 KiB weight
//     ring of kernels_chain.h, so the WEIGHTS GO STRAIGHT FROM L2 INTO VGPRs (a weight fragment has exactly one consumer wave: each
//     wave owns 16 of a tile's 128 output columns for all rows).  The stream is repacked in HALF stages of 8 KiB
//     ([128 output columns] x [32 k] = one MFMA k-chunk, [wave][lane][8 values]: one 1 KiB global_load_dwordx4 per wave IS the A
//     operand of MT MFMAs) in k-chunk-major order inside a tile group, and runs through an 8-deep register ring (32 VGPRs) of
//     ordinary loads whose s_waitcnt vmcnt(N) hipcc places itself.
//   * THE RESIDUAL ROWS DO NOT LIVE IN REGISTERS ACROSS THE GEMMS.  They are read where the FiLM + residual epilogue folds them
//     into the accumulators (the accumulators BECOME the rows), and stored back before the feed-forward block ("parked", 2 KiB per
//     row through L2) -- 80 rows x 512 columns are 80 registers per lane, which the FFN needs for its linear2 partials.
//   * EPILOGUE OPERANDS COME FROM LDS.  The ring's 80 KiB are gone, so the per-column operands of every epilogue -- out_proj /
//     linear2 biases, LayerNorm gamma / beta, the FiLM scale / shift rows of the (at most two) sequences a panel touches: 28 KiB --
//     are DMA'd next to the attention-output panel at kernel start.  In kernels_chain.h each of them is a global load at a
//     "turn-around" of the chain: ~11 dependent L2 round trips per POST kernel with the matrix pipe idle, 0.5-1 us each at B=8 and
//     several times that at B=32 (r03 phase stamps: FiLM 2.5 -> 7.9 us, LayerNorm + rotary 4 -> 17 us per 48-row panel).
//   * STORED GEMMs ([Q|K], V^T, Q) RUN IN PAIRS OF TILES, k-chunk-major: the panel fragments of a k-chunk are read once for both
//     tiles (8 waves each read every panel row: the LDS port is the second roof of these phases) and the pair is what the paired
//     column map writes as 64-byte runs anyway.
//
// Round 6 (docs/lab_notebook_r6.md sections 12-14), each bit-identical to what it replaces:
//   * THE FEED-FORWARD BLOCK HAS NO WORKGROUP BARRIER (CHAIN4_FFN_PIPE): per-chunk LDS counters, two chunk buffers (the second over LDS that is dead during the block), the
//     two waves of a SIMD one chunk apart so that one computes the GELU while the other issues MFMAs; the stream is packed per wave group.
//   * LAST = 2: final_layer (model/diffusion.py:397) inside the last decoder layer's POST kernel, as a split-operand exact island on the rows in registers.
//   * MODE = CHAIN_IN: input permute + input_projection (model/diffusion.py:345-346,364) + layer 0's PRE work in one launch.
//
// One workgroup = 8 waves (two 256-register waves per SIMD) per CU.  Host contract: d = 512, ff = 1024, FiLM present, rows_per_seq
// >= 16 * MT (a panel touches at most two sequences) and a multiple of 8 (staged V^T store); everything else takes kernels_chain.h.
#pragma once
#include "kernels_chain.h"

#pragma clang fp contract(off)

#define CHAIN4_PF 8            // half stages in flight per wave (register ring); every GEMM of a chain consumes a multiple of it
#ifndef CHAIN4_ASM_ALL
#define CHAIN4_ASM_ALL 0     // A/B builds: inline-asm panel-fragment reads in every tall kernel (default: the 80-row POST kernel only)
#endif
#ifndef CHAIN4_PF_MID
#define CHAIN4_PF_MID 16       // ... of the 48-row MID kernel, which has the registers (208 used): 27.2 -> 26.4 us at B=8 (profiles/r05_chain4_mid_ring16_ab.txt);
                               // at 80 rows a 16-deep ring spills 37-71 registers (38 -> 52 us), at 64 rows it is unmeasured: both keep 8
#endif
#ifndef CHAIN4_PF_POST3
#define CHAIN4_PF_POST3 8
#endif
#ifndef CHAIN4_PARK3
#define CHAIN4_PARK3 0         // 48-row POST kernel with parked rows (frees 48 registers, e.g. for a 16-deep ring: CHAIN4_PF_POST3=16)
#endif
#ifndef CHAIN4_ABL
#define CHAIN4_ABL 0           // scratch timing experiments (results wrong): 1 no weight loads after the ring is primed, 2 panel fragments read once per GEMM call, 4 the weight stream wraps inside its first 256 KiB (always L2-resident)
#endif
#define CHAIN4_HS_ELEMS 4096   // 128 output columns x 32 k: 8 KiB
#ifndef CHAIN4_SPREAD
#define CHAIN4_SPREAD 0        // stream layout: blocks of four half stages, a wave's four 1 KiB slices contiguous (4 KiB), the eight waves 4 KiB apart -- the eight requests of a
                               // half stage go to eight different 4 KiB blocks instead of one 8 KiB run (0: [half stage][wave][lane], A/B)
#endif

#ifndef CHAIN4_FFN_PIPE
#define CHAIN4_FFN_PIPE 1      // feed-forward block without workgroup barriers: per-chunk LDS counters, the two waves of a SIMD one phase apart (0: round-5 form, A/B)
#endif

template <int MT>
struct Chain4Lds {
  static constexpr int D = 512, BM = 16 * MT, AUX_F = 2560;
  // hidden chunk of the feed-forward block: 256 columns where the LDS has room (<= 64 rows), 128 at 80 rows.  256: linear1 computes
  // two hidden tiles per panel-fragment read (8 waves each read every panel row: with one tile the LDS port is linear1's roof) and the
  // block has half the chunk hand-offs.  The host packs the stream accordingly (a2p_lib_run.h: two POST streams).
  static constexpr int HC = MT <= 4 ? 256 : 128;
  // Epilogue block (fp32 offsets).  Operands that are dead once the feed-forward block starts come first: the SECOND hidden-chunk buffer
  // (CHAIN4_FFN_PIPE) lies over the LayerNorm partials + those operands (+ PAD_F floats where that is not enough), so 80 rows still fit.
  //   dead by the FFN: out_proj bias, norm-A gamma / beta, FiLM rows of the out_proj epilogue;  live: linear2 bias, norm-B gamma / beta, FiLM rows of the FFN epilogue
  // (FiLM blocks: [sequence A: scale 512 | shift 512][sequence B: scale 512 | shift 512])
  static constexpr int E_BIAS_O = 0, E_LNA_G = 512, E_LNA_B = 1024, E_FILM_O = 1536, E_DEAD_F = 3584;
  static constexpr int RED_F = 16 * BM;                                               // LayerNorm partials [2][8][BM] fp32
  static constexpr int PAD_F = (BM * HC) / 2 > RED_F + E_DEAD_F ? (BM * HC) / 2 - RED_F - E_DEAD_F : 0;
  static constexpr int E_BIAS_2 = E_DEAD_F + PAD_F, E_LNB_G = E_BIAS_2 + 512, E_LNB_B = E_BIAS_2 + 1024, E_FILM_F = E_BIAS_2 + 1536, EPI_F = E_BIAS_2 + 3584;
  static constexpr int FLAG_F = 32;                                                   // ready[16] | done[16] chunk counters
  // 16-bit elements: panelA [BM][512], panelH [BM][HC], LayerNorm partials, epilogue block [EPI_F] fp32, aux [AUX_F] fp32, counters
  static constexpr int ELEMS = BM * D + BM * HC + 2 * RED_F + 2 * EPI_F + 2 * AUX_F + 2 * FLAG_F;
  static_assert(ELEMS * 2 <= 160 * 1024, "panel too tall for the LDS");
};

// compile-time loop (the LDS reads below carry their offsets as instruction immediates)
template <class F, int... I>
__device__ __forceinline__ void chain4_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void chain4_static_for(F&& f) { chain4_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ds_read_b128 the compiler does not track: no s_waitcnt is inserted for the result -- chain4_lds_wait is the wait
template <int OFF>
__device__ __forceinline__ h16x8 chain4_lds_rd(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  h16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// wait until at most N LDS operations are outstanding; the fragments pass through the statement, so no consumer can be scheduled above it
template <int N, int MT>
__device__ __forceinline__ void chain4_lds_wait(h16x8 (&a)[MT]) {
  static_assert(MT >= 3 && MT <= 5, "panel height");
  if constexpr (MT == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N));
  else if constexpr (MT == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]) : "n"(N));
}

// one half stage of a packed stream: [wave 0..7][lane 0..63][8 k-values] -- lane (l15, g) of wave w gets
// W[col(w, l15)][k0 + g*8 .. +8], the weight operand of v_mfma_f32_16x16x32 for this k-chunk.  Column ownership as chain_pack_kernel
// with 8 waves (32-column group w >> 1, sub-tile w & 1, paired map for stored tiles).
// descs: [2][n] -- half stage i of waves 4-7 is descs[i], of waves 0-3 descs[n + i] (the two wave groups of a pipelined feed-forward block consume their
// linear1 / linear2 chunks in different orders; everywhere else the two entries are equal)
__global__ __launch_bounds__(256) void chain4_pack_kernel(const ChainPackDesc* __restrict__ descs, h16_t* __restrict__ dst, int n) {
  for (int q = threadIdx.x; q < 512; q += 256) {
    const int w = q >> 6, lane = q & 63, i = lane & 15, g = lane >> 4;
    const int64_t h = blockIdx.x;
    uint4* out = reinterpret_cast<uint4*>(dst) + (CHAIN4_SPREAD ? (h >> 2) * 2048 + w * 256 + (h & 3) * 64 + lane : h * 512 + q);
    const ChainPackDesc d = descs[(w < 4 ? n : 0) + blockIdx.x];
    const int w4 = w >> 1, J = w & 1, tile = d.row0 >> 7;
    const int row = d.omap ? (tile >> 1) * 256 + w4 * 64 + J * 32 + (tile & 1) * 16 + i : d.row0 + w4 * 32 + (i >> 2) * 8 + J * 4 + (i & 3);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < d.nrows) v = *reinterpret_cast<const uint4*>(d.W + (int64_t)row * d.ldw + d.k0 + g * 8);
    *out = v;
  }
}

template <int MT, int MODE, int LAST>
__device__ __forceinline__ void chain4_body(const ChainP& p, h16_t* const smem, const int m0) {
  constexpr int D = 512, NW = 8, CW = 16, BM = 16 * MT, NT = 4, KC = D / 32, FT = 8, HLD = Chain4Lds<MT>::HC, NH = HLD / 128, PF = (MODE == CHAIN_MID && MT == 3) ? CHAIN4_PF_MID : (MODE == CHAIN_POST && MT == 3) ? CHAIN4_PF_POST3 : CHAIN4_PF;
  using LY = Chain4Lds<MT>;
  constexpr int AUX_F = LY::AUX_F;
  constexpr int E4_BIAS_O = LY::E_BIAS_O, E4_BIAS_2 = LY::E_BIAS_2, E4_LNA_G = LY::E_LNA_G, E4_LNB_G = LY::E_LNB_G, E4_FILM_O = LY::E_FILM_O, E4_FILM_F = LY::E_FILM_F;
  h16_t* const panelA = smem;
  h16_t* const panelH = panelA + BM * D;
  float* const red = reinterpret_cast<float*>(panelH + BM * HLD);   // [2][8][BM] LayerNorm partial sums, one per wave
  float* const epi = red + LY::RED_F;                                 // [EPI_F] epilogue operands (Chain4Lds::E_*)
  float* const aux = epi + LY::EPI_F;                                 // [AUX_F] per-tile biases of the stored / FFN GEMMs
  uint32_t* const flags = reinterpret_cast<uint32_t*>(aux + AUX_F);   // [2][16] chunk counters of the feed-forward block (CHAIN4_FFN_PIPE)
  h16_t* const panelH1 = reinterpret_cast<h16_t*>(red);               // second hidden-chunk buffer: over red + the epilogue operands that are dead by then
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
  const int W4 = wid >> 1, J0 = wid & 1;
#ifdef A2P_STAMPS   // diagnostic build (scratch/phase_probe4.py): 100 MHz phase stamps into p.fin_out: workgroups 0 and 101, or (A2P_STAMPS == 2) waves 0 and 4 of workgroup 0 (one SIMD)
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if (A2P_STAMPS == 2 ? ((tid == 0 || tid == 256) && blockIdx.x == 0) : (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101))) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.fin_out) + ((A2P_STAMPS == 2 ? tid : blockIdx.x) ? 32 : 0);
      o[i] = wall_clock64();
      if (i == 0) o[30] = __builtin_readcyclecounter();   // shader cycles at the first / the latest stamp: the EFFECTIVE clock of the launch
      o[31] = __builtin_readcyclecounter();
    }
  };
#else
  auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
  stamp(0);

  // ---- weight stream: register ring of PF half stages --------------------------------------------------------------------
  uint32_t woff = CHAIN4_SPREAD ? (uint32_t)(wid * 4096 + lane * 16) : (uint32_t)(wid * 64 + lane) * 16;   // byte offset of this lane's 16 bytes of the next half stage to load
  h16x8 wr[PF];
  bool w_primed = false;
  auto w_issue = [&](int slot) __attribute__((always_inline)) {
    if ((CHAIN4_ABL & 1) && w_primed) { asm volatile("" : "+v"(wr[slot])); return; }
    if (CHAIN4_ABL & 4) wr[slot] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(p.stream) + (woff & 0x3ffffu));   // (every half stage from the first 256 KiB: L2 hits by construction)
    else wr[slot] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(p.stream) + woff);   // uniform base + 32-bit offset
    // the host pads the stream behind the last half stage.  (SPREAD: every GEMM starts on a multiple of the ring depth, so slot & 3 is the half stage's place in its block of four)
    woff += CHAIN4_SPREAD ? ((slot & 3) == 3 ? 32768u - 3072u : 1024u) : CHAIN4_HS_ELEMS * 2;
  };

  // ---- helpers -----------------------------------------------------------------------------------------------------------
  auto col_of = [&](int t) __attribute__((always_inline)) { return t * 128 + W4 * 32 + g * 8 + J0 * 4; };
  auto obase = [&](int t) __attribute__((always_inline)) { return (t >> 1) * 256 + W4 * 64 + J0 * 32 + (t & 1) * 16; };
  auto x_rbase = [&](int m, int tiled) __attribute__((always_inline)) -> uint32_t {   // float offset of this lane's 4 columns of tile 0 of row m
    const uint32_t a = (uint32_t)(((m >> 4) * (D / 16) + W4 * 2 + J0) * 256 + (g * 16 + (m & 15)) * 4);
    const uint32_t b = (uint32_t)(m * D + W4 * 32 + g * 8 + J0 * 4);
    return tiled ? a : b;
  };
  auto x_tstride = [&](int tiled) __attribute__((always_inline)) -> uint32_t { return tiled ? 2048u : 128u; };
  auto ld4 = [&](const float* base, uint32_t elem) __attribute__((always_inline)) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (elem << 2));
  };
  auto st4 = [&](float* base, uint32_t elem, f32x4 v) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + (elem << 2)) = v;
  };
  auto lds_off = [&](const void* q) __attribute__((always_inline)) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q; };
  int row_m[MT], fsel[MT], pos[MT];
  const int seqA = m0 / p.rows_per_seq;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < p.M ? m : p.M - 1;
    row_m[mt] = m;
    const int sq = m / p.rows_per_seq;
    fsel[mt] = sq != seqA ? 1024 : 0;                 // second sequence of the panel: its FiLM rows sit 1024 floats further
    pos[mt] = m - sq * p.rows_per_seq;
  }
  // fragment of k-chunk c (32 k-values) of panel rows mt*16 + l15: 16-byte piece (c*4 + g) ^ l15 of the row (XOR swizzle)
  uint32_t aswz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aswz[i] = (uint32_t)(((i * 4) ^ (g ^ l15)) << 4);
  const int pswz = (((W4 * 4 + g) ^ l15) << 3) | (J0 * 4);   // element offset inside a 128-column tile of this lane's 4 output columns

  // acc[t][mt] += P[:, 0 : 32*NKC] x (the next NKC*NTG half stages)^T for NTG tiles, k-chunk-major (half stage = c*NTG + t).
  // The panel fragments of k-chunk c+1 are meant to be read while the NTG*MT MFMAs of chunk c issue.  As ordinary loads hipcc
  // sinks every ds_read_b128 in front of its first MFMA with an s_waitcnt lgkmcnt(0) between them, whatever the source order or
  // the sched_group_barrier pattern says -- MT exposed LDS round trips per k-chunk, and with one tile per wave (linear1 of the
  // 80-row POST kernel) the fragment is re-read for every MFMA (profiles/r05_chain4_isa_census.txt: a third of that kernel's MFMAs
  // sat directly behind such a wait).  ASM_FRAGS: the reads are inline asm, issued one chunk ahead and waited for with a counted
  // lgkmcnt (the newest MT may still be in flight).  Measured on one box (profiles/r05_chain4_asm_frags_ab.txt): 80-row POST
  // -4.5 %, but the 48-row kernels and every MID kernel +2..3 % (the second wave of the SIMD already covered the round trips there and
  // the double-buffered fragments cost registers), so only that kernel takes it.  Ring slots are compile-time.
  // swap = false: D = C^T, lane holds 4 consecutive columns n of row m = l15; swap = true: D = C (transposed V^T store).
  auto gemm = [&](auto ntg_c, auto nkc_c, auto pld_c, auto& acc, const h16_t* P, bool swap) __attribute__((always_inline)) {
    constexpr int NTG = decltype(ntg_c)::value, NKC = decltype(nkc_c)::value, PLD = decltype(pld_c)::value;
    static_assert((NTG * NKC) % PF == 0 && PF % NTG == 0, "ring phase");
    constexpr bool ASM_FRAGS = CHAIN4_ASM_ALL || (MT == 5 && MODE == CHAIN_POST);
    if constexpr (ASM_FRAGS) {
      constexpr int RSTEP = 32 * PLD;   // bytes between the 16-row blocks of a panel
      const uint32_t rp = lds_off(P) + (uint32_t)(l15 * PLD * 2);
      uint32_t ab[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ab[i] = rp + aswz[i];
      h16x8 a[2][MT];
      auto rd_chunk = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int C = decltype(c_c)::value;
        chain4_static_for<MT>([&](auto mt_c) __attribute__((always_inline)) {
          constexpr int M = decltype(mt_c)::value, OFF = (C >> 2) * 256 + M * RSTEP;
          if constexpr (OFF < 65536) a[C & 1][M] = chain4_lds_rd<OFF>(ab[C & 3]);
          else a[C & 1][M] = chain4_lds_rd<OFF - 65536>(ab[C & 3] + 65536u);
        });
      };
      rd_chunk(std::integral_constant<int, 0>{});
      chain4_static_for<NKC>([&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value;
        if constexpr (c + 1 < NKC) {
          rd_chunk(std::integral_constant<int, c + 1>{});
          chain4_lds_wait<MT>(a[c & 1]);
        } else {
          chain4_lds_wait<0>(a[c & 1]);
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          const int slot = (c * NTG + t) % PF;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (swap) acc[t][mt] = A2P_MFMA16(a[c & 1][mt], wr[slot], acc[t][mt]);
            else acc[t][mt] = A2P_MFMA16(wr[slot], a[c & 1][mt], acc[t][mt]);
          }
          w_issue(slot);
        }
        // issue order of the chunk: the MT MFMAs of a tile, then the weight load that refills its ring slot
#pragma unroll
        for (int i = 0; i < NTG; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      });
    } else {
      const char* rp = reinterpret_cast<const char*>(P) + l15 * PLD * 2;
      constexpr int rstep = 32 * PLD;
      h16x8 a[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[0] + mt * rstep);
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        if (c + 1 < NKC) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (CHAIN4_ABL & 2) { a[(c + 1) & 1][mt] = a[c & 1][mt]; asm volatile("" : "+v"(a[(c + 1) & 1][mt])); }
            else a[(c + 1) & 1][mt] = *reinterpret_cast<const h16x8*>(rp + aswz[(c + 1) & 3] + ((c + 1) >> 2) * 256 + mt * rstep);
          }
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          const int slot = (c * NTG + t) % PF;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (swap) acc[t][mt] = A2P_MFMA16(a[c & 1][mt], wr[slot], acc[t][mt]);
            else acc[t][mt] = A2P_MFMA16(wr[slot], a[c & 1][mt], acc[t][mt]);
          }
          w_issue(slot);
        }
#pragma unroll
        for (int i = 0; i < NTG * MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (c + 1 < NKC && i < MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (i % MT == MT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
  };

  // ---- kernel start: attention-output panel + aux + epilogue operands by LDS-DMA, weight ring primed ---------------------
  {
    if constexpr (MODE != CHAIN_IN) {
      for (int r0 = wid; r0 < BM; r0 += NW) {   // one 1 KiB row per wave instruction
        int m = m0 + r0;
        m = m < p.M ? m : p.M - 1;
        m = (p.src_rows > 0 && m >= p.src_rows) ? m - p.src_rows : m;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ain + (int64_t)m * p.ld_ain + ((lane ^ (r0 & 15)) << 3)),
                                         (__attribute__((address_space(3))) void*)(panelA + r0 * D), 16, 0, 0);
      }
    }
    for (int kb = wid; kb < p.aux_kb; kb += NW) chain_glds16(p.aux + kb * 256 + lane * 4, aux + kb * 256);
    if (tid < LY::FLAG_F) flags[tid] = 0u;
    // epilogue block: 28 pieces of 1 KiB (256 floats); piece q -> source pointer + destination offset
    const int nseq = p.M / p.rows_per_seq;
    const int seqB = seqA + 1 < nseq ? seqA + 1 : seqA;
    for (int q = wid; q < 28; q += NW) {
      if (MODE == CHAIN_IN && (q < 8 || q >= 12)) continue;   // (norm-B gamma / beta only: no attention output, no FiLM in front of layer 0)
      const float* src;
      int dst;
      if (q < 12) {   // six 512-float vectors, two pieces each
        const int v = q >> 1, h = q & 1;
        const float* base = v == 0 ? p.bias_o : v == 1 ? p.bias_2 : v == 2 ? p.lnA_g : v == 3 ? p.lnA_b : v == 4 ? p.lnB_g : p.lnB_b;
        if (base == nullptr) base = p.bias_o;   // (MID: no bias_2 / lnB: never read)
        src = base + h * 256;
        dst = (v == 0 ? E4_BIAS_O : v == 1 ? E4_BIAS_2 : v == 2 ? E4_LNA_G : v == 3 ? E4_LNA_G + 512 : v == 4 ? E4_LNB_G : E4_LNB_G + 512) + h * 256;
      } else {        // FiLM rows: (film_o | film_f) x (sequence A | B) x (scale | shift) x 2 pieces
        const int r = q - 12, set = r >> 3, sq = (r >> 2) & 1, part = (r >> 1) & 1, h = r & 1;
        const float* film = set == 0 ? p.film_o : (p.film_f ? p.film_f : p.film_o);
        src = film + (int64_t)(sq ? seqB : seqA) * p.film_seq_stride + (part ? p.film_shift_off : 0) + h * 256;
        dst = (set == 0 ? E4_FILM_O : E4_FILM_F) + sq * 1024 + part * 512 + h * 256;
      }
      chain_glds16(src + lane * 4, epi + dst);
    }
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) w_issue(i);
  w_primed = true;
  if constexpr (MODE == CHAIN_IN) {
    // The noisy input x [B][256][T] (model/diffusion.py:345-346: permute to [T][256]) as the split operand panel [BM][hi 256 | lo 256] over the A panel: wave w takes
    // channels 32 w .. 32 w + 31, lane = panel row (consecutive frames: each load instruction reads BM contiguous floats of one channel).  What pack_input_split3_kernel
    // wrote to HBM for gemm_kernel to read back.
    for (int r0 = 0; r0 < BM; r0 += 64) {
      const int r = r0 + lane;
      if (r < BM) {
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        const int b = m / p.rows_per_seq, tpos = m - b * p.rows_per_seq;
        const float* src = p.xin + ((int64_t)b * p.xin_C + wid * 32) * p.rows_per_seq + tpos;
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = src[(int64_t)j * p.rows_per_seq];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          h16x8 hi, lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xv = v[jp * 8 + e];
            hi[e] = (h16_t)xv;
            lo[e] = (h16_t)(xv - (float)hi[e]);
          }
          const int q = wid * 4 + jp;   // 16-byte piece of the row's hi half; the lo half starts 32 pieces further; pieces are XOR-swizzled inside groups of 16 (the fragment reads' layout)
          *reinterpret_cast<h16x8*>(panelA + r * D + (((q & ~15) | ((q & 15) ^ (r & 15))) << 3)) = hi;
          *reinterpret_cast<h16x8*>(panelA + r * D + ((((q + 32) & ~15) | ((q & 15) ^ (r & 15))) << 3)) = lo;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces have landed for this wave (once per kernel: the ring's first loads too)
  chain_bar();                                        // ... and for every other wave
  stamp(1);

  // ---- epilogues ---------------------------------------------------------------------------------------------------------
  // FiLM affine + residual, IN PLACE: R[t][mt] = x_old + (scale + 1) * (R + bias) + shift   (transformer_modules.py:122-124,193).
  // bias / scale / shift from the LDS block; the old rows from global memory, one tile ahead of the arithmetic.
  auto film_res = [&](f32x4(&R)[NT][MT], int e_bias, int e_film, const float* xs, int tiled, bool use_src) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    uint32_t xb[MT];
    const uint32_t ts = x_tstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ms = (use_src && p.src_rows > 0 && row_m[mt] >= p.src_rows) ? row_m[mt] - p.src_rows : row_m[mt];
      xb[mt] = x_rbase(ms, tiled);
    }
    f32x4 xo[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xo[0][mt] = ld4(xs, xb[mt]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xo[(t + 1) & 1][mt] = ld4(xs, xb[mt] + (t + 1) * ts);
      }
      const f32x4 b = *reinterpret_cast<const f32x4*>(epi + e_bias + col_of(t));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(epi + e_film + fsel[mt] + col_of(t));
        const f32x4 sh = *reinterpret_cast<const f32x4*>(epi + e_film + fsel[mt] + 512 + col_of(t));
        const f32x4 y = R[t][mt] + b, s1 = sc + 1.0f;
        f32x4 xr = xo[t & 1][mt];
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[e] += fmaf(s1[e], y[e], sh[e]);
        R[t][mt] = xr;
        asm volatile("" : "+v"(R[t][mt]));   // the result is pinned here (hipcc otherwise sinks the arithmetic to the first use of the rows)
      }
      __builtin_amdgcn_sched_barrier(0);     // one tile at a time
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
  // normalised (optionally rotated, rotary_embedding_torch.py:46-66) rows -> 16-bit A panel.  gamma / beta from the LDS block; the
  // rotary entries (MT per tile, global: the panel-layout table, 256 contiguous bytes per lane group) one tile ahead of their use
  auto ln_write = [&](const f32x4(&R)[NT][MT], int e_gamma, auto rope_c) __attribute__((always_inline)) {
    constexpr bool ROPE = decltype(rope_c)::value;
    f32x4 cs[2][ROPE ? MT : 1];
    auto load_cs = [&](int t) __attribute__((always_inline)) {
      if constexpr (ROPE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          cs[t & 1][mt] = ld4(reinterpret_cast<const float*>(p.cst), ((uint32_t)(col_of(t) >> 2) * (uint32_t)p.cs_npos + (uint32_t)pos[mt]) << 2);
      }
    };
    load_cs(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_barrier(0);     // at most two tiles' rotary entries in flight
      if (t + 1 < NT) load_cs(t + 1);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(epi + e_gamma + col_of(t));
      const f32x4 be = *reinterpret_cast<const f32x4*>(epi + e_gamma + 512 + col_of(t));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float rs = ln_rstd[mt], nm = -ln_mean[mt] * rs;
        float v0 = fmaf(fmaf(R[t][mt][0], rs, nm), ga[0], be[0]);
        float v1 = fmaf(fmaf(R[t][mt][1], rs, nm), ga[1], be[1]);
        float v2 = fmaf(fmaf(R[t][mt][2], rs, nm), ga[2], be[2]);
        float v3 = fmaf(fmaf(R[t][mt][3], rs, nm), ga[3], be[3]);
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
  auto load_x = [&](f32x4(&R)[NT][MT], const float* xs, int tiled) __attribute__((always_inline)) {
    const uint32_t ts = x_tstride(tiled);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const uint32_t xb = x_rbase(row_m[mt], tiled);
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
  // D-deep GEMM over NGRP groups of FOUR output tiles with 16-bit stores (kernels_chain.h gemm_store, 8-wave forms): the two tile
  // pairs of a group leave as 16 rows x 64 contiguous bytes per instruction through a wave-private slice of the idle hidden-chunk
  // buffer (paired column map), V^T tiles as 16-byte pieces of 8 consecutive frames.  Groups of four, not pairs: every burst of stores
  // sits in front of the ring's next loads in the (in-order) memory queue, i.e. a store acknowledgement -- 1-3 us at B=32 -- is exposed
  // per burst whenever the ring (8 half stages = 0.6 us of MFMAs) runs dry behind it: first version, pairs, [Q|K] phase 27.8 us for a
  // 10 us GEMM (profiles/r05_tall_chain_phase_stamps_v0.txt).  Fully unrolled: across a loop back-edge hipcc waits for ALL ring loads.
  auto gemm_store = [&](auto ngrp_c, const float* bias_lds, h16_t* out, int64_t ldo, auto transposed_c) __attribute__((always_inline)) {
    constexpr int NGRP = decltype(ngrp_c)::value;
    constexpr bool TR = decltype(transposed_c)::value;
    constexpr int VP = (CW * BM / 8 + 63) / 64;
    h16_t* const stg = panelH + wid * (CW * BM);
    uint32_t voff[VP];
    bool vok[VP];
    if constexpr (TR) {
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int q = lane + 64 * i, c = q / (BM / 8), m = m0 + (q % (BM / 8)) * 8;
        const int sq = m / p.rows_per_seq;
        vok[i] = q < CW * BM / 8 && m < p.M;
        voff[i] = (uint32_t)sq * (uint32_t)p.vt_seq_stride + (uint32_t)(m - sq * p.rows_per_seq) + (uint32_t)c * (uint32_t)ldo;
      }
    }
#pragma unroll
    for (int gr = 0; gr < NGRP; ++gr) {
      f32x4 acc[4][MT];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int t = 4 * gr + h;
        if constexpr (!TR) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias_lds + obase(t) + g * 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[h][mt] = b;
        } else {
          const float b = bias_lds[obase(t) + l15];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[h][mt] = f32x4{b, b, b, b};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      gemm(std::integral_constant<int, 4>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelA, TR);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!TR) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v0 = acc[2 * pr][mt], v1 = acc[2 * pr + 1][mt];
            const h16x4 lo = {(h16_t)v0[0], (h16_t)v0[1], (h16_t)v0[2], (h16_t)v0[3]};
            const h16x4 hi = {(h16_t)v1[0], (h16_t)v1[1], (h16_t)v1[2], (h16_t)v1[3]};
            const int hsw = (l15 >> 2) & 1;   // half-row swizzle of the [16][32] staging tile (kernels_chain.h)
            asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" ::"v"(lds_off(stg + l15 * 32 + hsw * 16 + g * 4)),
                         "v"(lds_off(stg + l15 * 32 + (hsw ^ 1) * 16 + g * 4)), "v"(lo), "v"(hi)
                         : "memory");
            h16x8 w;
            const int prow = lane >> 2, pp = lane & 3;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(w)
                         : "v"(lds_off(stg + prow * 32 + (((pp >> 1) ^ ((prow >> 2) & 1)) * 2 + (pp & 1)) * 8))
                         : "memory");
            const int m = m0 + mt * 16 + (lane >> 2);
            if (m < p.M)
              *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + (((uint32_t)m * (uint32_t)ldo + (uint32_t)(obase(4 * gr + 2 * pr) + (lane & 3) * 8)) << 1)) = w;
          }
      } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[h][mt];
            const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            asm volatile("ds_write_b64 %0, %1" ::"v"(lds_off(stg + l15 * BM + mt * 16 + g * 4)), "v"(o) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < VP; ++i) {
            h16x8 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(stg + (lane + 64 * i) * 8)) : "memory");
            if (vok[i]) *reinterpret_cast<h16x8*>(reinterpret_cast<char*>(out) + ((voff[i] + (uint32_t)obase(4 * gr + h) * (uint32_t)ldo) << 1)) = v;
          }
        }
      }
    }
  };

  // =========================================================================================================================
  const std::false_type F{};
  const std::true_type Tt{};
  f32x4 R[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) R[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (MODE == CHAIN_IN) {
    // input_projection (model/diffusion.py:364) as a split-operand exact island: x = hi + lo, rows = hi W_hi^T + lo W_hi^T + hi W_lo^T + bias, in gemm_kernel's k order
    // (the stream carries [W_hi | W_hi | W_lo], 256 k each, four tiles k-chunk-major), accumulators from zero, the bias last: the bits of pack_input_split3 + gemm_kernel.
    gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, KC / 2>{}, std::integral_constant<int, D>{}, R, panelA, false);
    __builtin_amdgcn_sched_barrier(0);
    gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, KC / 2>{}, std::integral_constant<int, D>{}, R, panelA + D / 2, false);
    __builtin_amdgcn_sched_barrier(0);
    gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, KC / 2>{}, std::integral_constant<int, D>{}, R, panelA, false);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(aux + col_of(t));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) R[t][mt] += b;
    }
    chain_bar();   // every wave is done reading the input panel (the LayerNorm below rewrites it)
    stamp(2);
  } else {
    gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, R, panelA, false);   // out_proj of the attention that produced `ain`
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);
    film_res(R, E4_BIAS_O, E4_FILM_O, p.xsrc ? p.xsrc : p.x, p.x_in_tiled, true);
    stamp(3);
  }
  // POST kernels that park their rows around the feed-forward block: the store leaves here, in front of the LayerNorm (see the second park below for why)
  constexpr bool PARK_FFN = MODE == CHAIN_POST && (MT >= 4 || CHAIN4_PARK3);
  if constexpr (PARK_FFN) {
    if (!p.x_in_tiled) chain_bar();   // (parked in the tiled layout: with a row-major predecessor layout -- A/B only -- other lanes' unread bytes lie under the store)
    store_x(R, 1);
  }
  if constexpr (MODE != CHAIN_IN) {
    ln_stats(R);   // (its barriers also order the panel rewrite behind every wave's out_proj reads)
    stamp(4);
  }
  if constexpr (MODE == CHAIN_MID) {
    ln_write(R, E4_LNA_G, Tt);
    stamp(5);
    gemm_store(std::integral_constant<int, 1>{}, aux, p.q_out, p.ld_q, F);
    stamp(6);
    store_x(R, p.x_out_tiled);
    stamp(7);
  } else {
    // 64 / 80-row panels: the rows are PARKED (stored, re-read) around the feed-forward block and again around the [Q|K] GEMM -- 80 registers
    // per lane that the linear2 partials / the four-tile groups need.  At 48 rows they stay in registers from load to the final
    // store (X), as in kernels_chain.h: no extra traffic, and no store in front of a GEMM's weight loads.
    constexpr bool PARK = MT >= 4 || CHAIN4_PARK3;   // (64 rows without parking: 91 registers spilled)
    if constexpr (MODE != CHAIN_IN) {   // (CHAIN_IN: the rows just computed go straight to layer 0's PRE work below)
    ln_write(R, E4_LNA_G, F);
    [[maybe_unused]] f32x4 X[PARK ? 1 : NT][PARK ? 1 : MT];
    // (parked in the tiled layout whatever the final layout is: a 16-row block occupies the same bytes in both, and the workgroup owns
    // whole blocks -- the last layer writes its rows back row-major)
    if constexpr (!PARK) {   // (PARK: stored in front of the LayerNorm, above)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) X[t][mt] = R[t][mt];
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(5);
    // Feed forward, split-K over the hidden chunks of HLD columns: linear1 chunk (NH tiles, k-chunk-major) -> GELU -> LDS -> linear2
    // partial.  Fully unrolled: across a loop back-edge hipcc waits for ALL ring loads (one exposed L2 round trip per iteration).
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) R[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if CHAIN4_FFN_PIPE
    // No workgroup barrier inside the block.  Per hidden chunk two LDS counters: ready[c] = waves whose GELU'd columns of chunk c are in the buffer,
    // done[c] = waves that have read chunk c for their linear2 partial.  Two chunk buffers (c & 1).  The waves w and w + 4 share a SIMD; waves 0-3 run
    // their linear2 partial ONE CHUNK LATE (linear1(c+1) -> GELU(c+1) -> linear2(c)), waves 4-7 in the natural order (linear1(c) -> GELU(c) -> linear2(c)):
    // after the first chunk one wave of every SIMD is in its GELU (vector unit) while the other issues MFMAs, instead of all eight computing the GELU
    // between two barriers with the matrix pipe idle (round 5: 1.6-2.9 us per chunk of 256 at 48 rows, 19 of 48 us of the block at 80 rows).  Same products,
    // same accumulation order (chunks 0, 1, 2 .. into R): bit-identical.  The weight stream is packed per wave group in ITS consumption order (a2p_lib_run.h).
    {
      constexpr int NC = FT / NH;
      const int skew = wid < 4 ? 1 : 0;
      const uint32_t fl = lds_off(flags);
      auto arrive = [&](uint32_t byte_off) __attribute__((always_inline)) {   // this wave's LDS traffic so far is complete; count it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(fl + byte_off), "v"(1u) : "memory");
      };
      auto wait_all = [&](uint32_t byte_off) __attribute__((always_inline)) {   // until all eight waves have counted
        uint32_t seen;
        asm volatile(
            ".Lc4w_%=:\n\t"
            "ds_read_b32 %0, %1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_u32 vcc, 8, %0\n\t"
            "s_cbranch_vccz .Lc4d_%=\n\t"
            "s_sleep 1\n\t"
            "s_branch .Lc4w_%=\n\t"
            ".Lc4d_%=:"
            : "=&v"(seen)
            : "v"(fl + byte_off)
            : "vcc", "memory");
      };
#pragma unroll
      for (int i = 0; i <= NC; ++i) {
        if (i < NC) {
          f32x4 acc[NH][MT];
#pragma unroll
          for (int tt = 0; tt < NH; ++tt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(aux + (i * NH + tt) * 128 + W4 * 32 + g * 8 + J0 * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[tt][mt] = b;
          }
          __builtin_amdgcn_sched_barrier(0);
          gemm(std::integral_constant<int, NH>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelA, false);
          __builtin_amdgcn_sched_barrier(0);
          if (13 + 3 * i < 28) stamp(13 + 3 * i);       // (stamped builds: linear1 of this hidden chunk done)
          if (i >= 2) wait_all(64u + 4u * (i - 2));   // every wave has read chunk i - 2: its buffer is free
          h16_t* const Hw = (i & 1) ? panelH1 : panelH;
#pragma unroll
          for (int tt = 0; tt < NH; ++tt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 v = acc[tt][mt];
              *reinterpret_cast<h16x4*>(Hw + (mt * 16 + l15) * HLD + tt * 128 + pswz) =
                  h16x4{(h16_t)act_gelu_fast(v[0]), (h16_t)act_gelu_fast(v[1]), (h16_t)act_gelu_fast(v[2]), (h16_t)act_gelu_fast(v[3])};
            }
          arrive(4u * i);
          if (14 + 3 * i < 29) stamp(14 + 3 * i);       // (GELU written and counted)
        }
        const bool do2 = i == 0 ? skew == 0 : i == NC ? skew == 1 : true;
        if (do2) {
          const int c = i - skew;
          wait_all(4u * c);                             // the hidden chunk is complete
          __builtin_amdgcn_sched_barrier(0);
          gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, HLD / 32>{}, std::integral_constant<int, HLD>{}, R, (c & 1) ? panelH1 : panelH, false);
          __builtin_amdgcn_sched_barrier(0);
          arrive(64u + 4u * c);
        }
        if (i < NC && 15 + 3 * i < 30) stamp(15 + 3 * i);     // (linear2 partial of chunk i - skew, where this wave has one in this round)
      }
      wait_all(64u + 4u * (NC - 1));   // the second chunk buffer lies over the LayerNorm partials: nobody reads it any more
    }
#else
#pragma unroll
    for (int h = 0; h < FT / NH; ++h) {
      f32x4 acc[NH][MT];
#pragma unroll
      for (int tt = 0; tt < NH; ++tt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(aux + (h * NH + tt) * 128 + W4 * 32 + g * 8 + J0 * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[tt][mt] = b;
      }
      __builtin_amdgcn_sched_barrier(0);
      gemm(std::integral_constant<int, NH>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelA, false);
      __builtin_amdgcn_sched_barrier(0);
      if (13 + 3 * h < 28) stamp(13 + 3 * h);       // (stamped builds: linear1 of this hidden chunk done in wave 0)
      if (h > 0) chain_bar();   // every wave finished the linear2 partial of the previous chunk
#pragma unroll
      for (int tt = 0; tt < NH; ++tt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[tt][mt];
          *reinterpret_cast<h16x4*>(panelH + (mt * 16 + l15) * HLD + tt * 128 + pswz) =
              h16x4{(h16_t)act_gelu_fast(v[0]), (h16_t)act_gelu_fast(v[1]), (h16_t)act_gelu_fast(v[2]), (h16_t)act_gelu_fast(v[3])};
        }
      chain_bar();              // the hidden chunk is complete
      if (14 + 3 * h < 29) stamp(14 + 3 * h);       // (GELU + both barriers)
      __builtin_amdgcn_sched_barrier(0);
      gemm(std::integral_constant<int, NT>{}, std::integral_constant<int, HLD / 32>{}, std::integral_constant<int, HLD>{}, R, panelH, false);
      __builtin_amdgcn_sched_barrier(0);
      if (15 + 3 * h < 30) stamp(15 + 3 * h);       // (linear2 partial)
    }
#endif
    stamp(6);
    if constexpr (PARK) {
      // the parked rows come back from where store_x left them (same workgroup, same lanes: program order makes them visible)
      film_res(R, E4_BIAS_2, E4_FILM_F, p.x, 1, false);
    } else {   // FiLM affine + residual on the register rows (film_res's arithmetic, operands from the LDS block)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(epi + E4_BIAS_2 + col_of(t));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(epi + E4_FILM_F + fsel[mt] + col_of(t));
          const f32x4 sh = *reinterpret_cast<const f32x4*>(epi + E4_FILM_F + fsel[mt] + 512 + col_of(t));
          const f32x4 y = R[t][mt] + b, s1 = sc + 1.0f;
          f32x4 xr = X[t][mt];
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[e] += fmaf(s1[e], y[e], sh[e]);
          R[t][mt] = xr;
        }
      }
    }
    stamp(7);
    if constexpr (LAST == 2) {
      // Last decoder layer, <= 64 rows: final_layer (model/diffusion.py:397) as a split-operand exact island on the rows in registers -- out = hi W_hi^T + lo W_hi^T + hi W_lo^T
      // with x = hi + lo (16-bit pieces), fp32 accumulation: what split3_kernel + gemm_kernel compute behind the non-fused kernel (19.7 MB of fp32 rows out, 29.5 MB of
      // split rows out and in again at B=8: 12.5 + ~30 us per step), without the round trip.  hi panel over the A panel, lo panel over the two hidden-chunk buffers (contiguous,
      // [BM][512] where the chunk is 256 wide); the stream carries [W_hi | W_hi | W_lo] as three 256 x 512 GEMMs in pairs of tiles.  The residual stream is NOT written back.
      h16_t* const panelLo = panelH;   // the two chunk buffers, contiguous: [BM][512] at <= 64 rows, [BM][256] (half of K at a time) at 80 rows
      constexpr bool LO_HALVES = 2 * HLD != D;
      static_assert(2 * HLD == D || 4 * HLD == D, "chunk buffers");
      constexpr int LLD = LO_HALVES ? D / 2 : D;
      // (every wave has passed wait_all(done[last]): nobody reads the chunk buffers; every wave counted ready[last] before that: nobody reads the A panel)
      auto split_rows = [&](auto hi_c, int t_lo0, int t_lo1) __attribute__((always_inline)) {   // hi pieces of every tile (once), lo pieces of tiles [t_lo0, t_lo1)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = R[t][mt];
            const h16x4 hi = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            if constexpr (decltype(hi_c)::value) *reinterpret_cast<h16x4*>(panelA + (mt * 16 + l15) * D + t * 128 + pswz) = hi;
            if (t >= t_lo0 && t < t_lo1) {
              const h16x4 lo = {(h16_t)(v[0] - (float)hi[0]), (h16_t)(v[1] - (float)hi[1]), (h16_t)(v[2] - (float)hi[2]), (h16_t)(v[3] - (float)hi[3])};
              *reinterpret_cast<h16x4*>(panelLo + (mt * 16 + l15) * LLD + (LO_HALVES ? (t & 1) : t) * 128 + pswz) = lo;
            }
          }
      };
      split_rows(Tt, 0, LO_HALVES ? 2 : NT);
      chain_bar();
      // (accumulators from zero, k in gemm_kernel's order, the bias added last: the same bits as the launches this replaces -- the kernel family of a forward,
      // which decides whether this kernel or kernels_chain.h + split3_kernel + gemm_kernel runs, stays invisible in the results: tests/test_hip_parity.py)
      f32x4 acc[2][MT];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[tt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_sched_barrier(0);
      gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelA, false);    // hi x W_hi
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LO_HALVES) {
        gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelLo, false);   // lo x W_hi
      } else {   // 80 rows: the lo pieces of k < 256, then of k >= 256, through the same 40 KiB (the stream's k order is the same either way: one group of two tiles)
        gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, KC / 2>{}, std::integral_constant<int, LLD>{}, acc, panelLo, false);
        __builtin_amdgcn_sched_barrier(0);
        chain_bar();
        split_rows(F, 2, 4);
        chain_bar();
        __builtin_amdgcn_sched_barrier(0);
        gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, KC / 2>{}, std::integral_constant<int, LLD>{}, acc, panelLo, false);
      }
      __builtin_amdgcn_sched_barrier(0);
      gemm(std::integral_constant<int, 2>{}, std::integral_constant<int, KC>{}, std::integral_constant<int, D>{}, acc, panelA, false);    // hi x W_lo
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + mt * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) st4(p.fin_out, (uint32_t)m * (uint32_t)p.ld_fin + (uint32_t)col_of(tt), acc[tt][mt] + *reinterpret_cast<const f32x4*>(aux + FT * 128 + col_of(tt)));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    if constexpr (LAST == 1) {   // last decoder layer (final_layer runs on its own exact-island path): the rows go back in the caller's layout.
      // Its own instantiation, not a run-time branch on p.has_next: with both continuations in one function hipcc spilled 70 registers of
      // the 80-row kernel (17 without the branch) and its POST launches at B=32 went from 232 to 263 us.
      // Parked rows were re-read in the TILED layout just now and the final layout is row-major: inside a 16-row block the two
      // layouts put different lanes' data on the same bytes, so no wave may store before every wave has its parked rows back
      // (first version without this barrier: wrong rows in every parked case, tests/test_hip_round5.py)
      if constexpr (PARK) chain_bar();
      store_x(R, p.x_out_tiled);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    }   // MODE != CHAIN_IN
    // next layer's PRE work: norm1 -> rotary -> [Q|K] ; norm1 -> V^T     (aux: bias_qk right behind bias_1, then bias_v)
    // Parked kernels: the finished rows leave HERE, in front of the LayerNorm, not behind it (their registers are free during the [Q|K] GEMM either way).
    // Every store sits in front of the weight ring's next loads in the in-order vmcnt queue, and this one is 2 KiB per row from every workgroup at once:
    // behind the LayerNorm its acknowledgement was exposed at the start of the [Q|K] GEMM (80 rows, B=32: that phase took 28-30 us against 12 us of
    // weight stream); here it drains during the two LayerNorm barriers and the panel rewrite, when the ring is full and nothing waits on it.
    // (The rows are overwritten in the layout they were parked in, lane for lane; a row-major successor layout -- A2P_CHAIN_X_ROWMAJOR, A/B only --
    // puts other lanes' parked bytes under the store: barrier first.)
    if constexpr (PARK) {
      if (!p.x_out_tiled) chain_bar();
      store_x(R, p.x_out_tiled);
    }
    ln_stats(R);
    ln_write(R, E4_LNB_G, Tt);
    stamp(8);
    __builtin_amdgcn_sched_barrier(0);
    stamp(9);
    gemm_store(std::integral_constant<int, 2>{}, aux + FT * 128, p.qk_out, p.ld_qk, F);
    chain_bar();                          // every wave is done reading the rotated panel
    stamp(10);
    if constexpr (PARK) load_x(R, p.x, p.x_out_tiled);   // back from where store_x left them
    ln_write(R, E4_LNB_G, F);
    stamp(11);
    gemm_store(std::integral_constant<int, 1>{}, aux + FT * 128 + 2 * D, p.vt_out, p.ld_vt, Tt);
    if constexpr (!PARK) store_x(R, p.x_out_tiled);  // last: loads return in issue order behind stores
    stamp(12);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's run-ahead loads target this wave's registers
}

template <int MT, int MODE, int LAST = 0>   // LAST: the POST kernel behind the last decoder layer (ChainP::has_next == 0); 2 = with final_layer as a split-operand island (ChainP::fin_x3)
__global__ __launch_bounds__(512, 2) void chain4_kernel(const ChainP p) {
  static_assert(!LAST || MODE == CHAIN_POST, "only POST has a last-layer form");
  __shared__ __attribute__((aligned(16))) h16_t smem[Chain4Lds<MT>::ELEMS];
  chain4_body<MT, MODE, LAST>(p, smem, blockIdx.x * (16 * MT));
}
#pragma clang fp contract(fast)
