// Row-panel "chain" kernels of the FiLM decoder layer (bf16 throughput mode).
//
// One workgroup owns BM = 16*MT rows of the fp32 residual stream for a whole CHAIN of dependent
// row-local operations (FiLMTransformerDecoderLayer.forward, transformer_modules.py:178-267):
//
//   MODE_PRE  : x -> LayerNorm(norm1) -> rotary -> [Q|K] projection, V projection (transposed store)
//   MODE_MID  : attention output -> out_proj -> FiLM affine + residual -> LayerNorm -> rotary -> Q projection
//   MODE_POST : attention output -> out_proj -> FiLM + residual -> LayerNorm(norm3) -> linear1 -> GELU ->
//               linear2 -> FiLM + residual [-> next layer's MODE_PRE work]
//
// instead of one launch (and one HBM round trip of the [rows, d] activation) per GEMM / LayerNorm:
// 5 launches per decoder layer instead of 12, the FFN hidden activation never leaves the CU, and the
// LayerNorm statistics are taken on the fp32 rows while they are still in registers.
//
// Data movement per workgroup (4 waves, one per SIMD):
//   * A operand: a bf16 [BM][d] panel in LDS (attention output fetched with global_load_lds, or the
//     LayerNorm output written from registers); 16-byte chunks XOR-swizzled by (row & 15) so the
//     ds_read_b128 fragment reads of 16 consecutive rows hit 16 different bank groups.
//   * W operand: all weight matrices of a chain are PRE-PACKED (a2p_finalize_weights, chain_pack_kernel)
//     into one contiguous stream of 16 KiB stages in exactly the order the chain consumes them; stage =
//     [wave 0..3][32 out-cols][64 k], already in the swizzled LDS image.  Wave w copies only ITS 4 KiB
//     slice (4 x global_load_lds of 1 KiB, perfectly sequential in HBM/L2) into a wave-private ring of NS
//     slots and waits with a counted s_waitcnt vmcnt -- no workgroup barrier in the GEMM loops, the DMA
//     queue is never drained, and the prefetch runs ahead across tiles, GEMMs and epilogues alike.
//   * No global load is issued while the stream is in flight except at the two "turn-around" points of a
//     chain (FiLM + LayerNorm after out_proj / after linear2), where the FiLM, LayerNorm and rotary
//     operands are fetched together; per-tile biases sit in LDS (DMA'd at kernel start).
//   * The residual rows live in registers (fp32) from load to final store; FFN hidden activations go
//     GELU -> bf16 -> a [BM][128] LDS chunk that linear2 consumes immediately (split-K over the 8 chunks).
//
// Wave w owns output columns [tile*128 + w*32, +32) of every 128-column tile, all BM rows: MT x 2 MFMA
// 16x16x32 fragments per k-chunk.  Accumulator layout (operands swapped, D = C^T): lane (l15 = lane & 15,
// g = lane >> 4) holds C[m = mt*16 + l15][n = ... + j*16 + g*4 + r], r = 0..3.
//
// Round 2 on top of that (each one bit-identical to the form it replaced; docs/lab_notebook_r1_r4.md section 4.1 has the measurements):
//   * two workgroup shapes, 4 waves x 512 registers or 8 waves x 256 (NW), same bits, chosen per box at run time;
//   * out_proj and the linear2 partials run as k-major GROUP GEMMs over all output tiles (gemm_group): the panel fragments of a
//     k-step are read once for the group, one pipeline ramp per group;
//   * every store writes >= 64 contiguous bytes per row: V^T tiles and (8 waves) Q/K/V tile pairs are transposed through a
//     wave-private slice of the idle hidden-chunk buffer, the 8-wave shape owns its columns under a paired map;
//   * the fp32 residual rows are tiled per 16-row block between chain kernels (one 1 KiB run per load / store instruction) and
//     stored last; the rotary table comes in the matching panel layout;
//   * chain_kernel_mix launches two panel heights at once so that large forwards fill whole rounds of the 256 CUs.
#pragma once
#include "a2p_common.h"

// Every instantiation of chain_kernel (panel height MT, 4 or 8 waves) must produce the SAME bits: the host picks MT from the
// row count and the workgroup shape from in-situ timings.  hipcc's default -ffp-contract=fast lets the compiler fuse a*b+c into
// an fma per instantiation as its scheduler sees fit; one instantiation (d=256, MT=2, 8 waves) fused the LayerNorm / GELU
// epilogue differently from the others -- 1-ulp fp32 differences that flip a bf16 rounding in ~1e-5 of the panel elements and
// then change a whole output row by ~2e-2 (round-1 open item, localised with scratch/chain_dbg.hip).  Contraction is therefore
// OFF in this header and every fused multiply-add is written as an explicit fmaf.
#pragma clang fp contract(off)

enum { CHAIN_PRE = 0, CHAIN_MID = 1, CHAIN_POST = 2, CHAIN_MIDPOST = 3, CHAIN_IN = 4 };   // CHAIN_IN (kernels_chain4.h only): input_projection + layer 0's PRE work
// CHAIN_MIDPOST (round 4, body model): MID work of the audio cross attention's output -> the KEYFRAME cross attention
// (multihead_attn2: <= 32 keys per sequence, transformer_modules.py:206-215) inside the kernel, on the query panel in LDS -> POST
// work.  One launch where MID2 | attention | POST were three: the residual rows stay in registers, the query never leaves the CU,
// and the 1280-workgroup attention launch over 20 keys (9 us of dispatch for 2 us of work) disappears.
// An in-kernel L2 look-ahead of the stream (one 4-byte-per-lane LDS-DMA per wave per stage touching the slice 8-16 stages
// ahead of the DMA head) was measured and rejected: +30 % kernel time warm AND cold -- the L2 request count per line, not
// the bytes, is what the extra instruction doubles (scratch/chain_bench, docs/lab_notebook_r1_r4.md section 4).
#define CHAIN_STREAM_PAD 8  // stages the host appends to a stream: the DMA runs up to NS-1 (<= 5) stages past the end
#define CHAIN_STAGE_ELEMS 8192  // 128 out-cols x 64 k bf16 = 16 KiB; 2048 elements (4 KiB) per wave

struct ChainP {
  int M, rows_per_seq, has_next, aux_kb;
  int n_tall;             // chain_kernel_mix: number of leading workgroups that take the taller panel
  int src_rows;           // > 0: the residual rows and the attention-output panel of row m are READ at row m - src_rows when m >= src_rows
                          // (layer 0 under classifier-free guidance: both halves of the batch enter with the same x and the same
                          // self attention, so the first PRE kernel and the first self attention run on one half only)
  const float* xsrc;      // residual rows are read from here instead of x when non-null: with src_rows the read rows of one
                          // workgroup are the WRITTEN rows of another, so the source must be a buffer this launch does not write
  float* x;               // fp32 residual stream [M][D], updated in place
  // Layout of the residual rows between chain kernels (x_in_tiled: as read, x_out_tiled: as written): per block of 16 rows the
  // D/16 "chunks" (tile t, 32-column group W4, 4-column half jj) of 1 KiB each, chunk = [g = 0..3][row & 15][4 floats] -- i.e.
  // exactly one load / store instruction of a wave (lane = g*16 + row): 1 KiB contiguous instead of 64 16-byte segments 2 KiB
  // apart.  A 16-row block occupies the same bytes in both layouts and a workgroup owns whole blocks, so the buffer can change
  // layout in place; the first kernel of a forward reads row-major (input projection), the last one writes row-major.
  int x_in_tiled, x_out_tiled;
  const h16_t* stream;   // packed weight stream of this chain
  const h16_t* stream4w; // POST only: the kernels_chain4.h stream with 256-column hidden chunks (panels of <= 64 rows)
  const h16_t* stream4;  // the same chain's stream in the half-stage register layout of kernels_chain4.h (NULL: not built for this chain)
  const float* aux;       // per-tile biases of this chain, aux_kb KiB: POST [bias_1 | bias_qk' | bias_v'], MID [bias_q], PRE [bias_qk | bias_v]
  // MID / POST: attention output panel
  const h16_t* ain;
  int64_t ld_ain;
  // out_proj epilogue
  const float* bias_o;
  const float* film_o;  // scale at film_o[seq*film_seq_stride + n], shift at + film_shift_off; NULL = plain residual
  int64_t film_seq_stride;
  int film_shift_off;
  // LayerNorm after out_proj (norm2 / norm2a / norm3)
  const float* lnA_g;
  const float* lnA_b;
  // MID: query projection of the following cross attention
  h16_t* q_out;
  int64_t ld_q;
  // POST: feed forward
  const float* bias_2;
  const float* film_f;
  // PRE work (MODE_PRE, or the tail of MODE_POST when has_next): norm1 -> rotary -> [Q|K], V^T
  const float* lnB_g;
  const float* lnB_b;
  h16_t* qk_out;
  int64_t ld_qk;
  h16_t* vt_out;
  int64_t vt_seq_stride, ld_vt;
  const f32x4* cst;  // rotary table in the panel layout [D/4][cs_npos]: (cos, sin) x 2 of 4 consecutive columns (rope_table_t_kernel)
  int cs_npos;
  // CHAIN_MIDPOST: the second sublayer (multihead_attn2 on the keyframe tokens) between the MID and POST work
  const float* bias_q2;        // in_proj_bias rows [0, D) of multihead_attn2 (its weight rows are the stream's second group GEMM)
  const float* bias_o2;        // its out_proj bias / FiLM (strides as film_o)
  const float* film_o2;
  const float* lnC_g;          // norm3 (lnA_* is norm2a in this mode)
  const float* lnC_b;
  const h16_t* k2;             // keyframe K cache [slot][keys][ld_k2] (+ layer offset), 16-bit
  const h16_t* vt2;            // keyframe V^T cache [slot][D rows][ld_vt2]
  int64_t k2_slot_stride, ld_k2, vt2_slot_stride, ld_vt2;
  const int* kv2_slots;        // per-sequence slot table, or the rule below (AttnP::slot_rule)
  int kv2_rule, kv2_b, n_key2; // n_key2 <= 32
  float scale2;                // 1 / sqrt(head_dim)
  int* stat_max;               // a2p_attention_logit_max (kernels_attn.h), may be NULL
  // has_next == 2 (last decoder layer): final_layer (model/diffusion.py:397) instead of the next layer's PRE work;
  // fp32 rows out[m][0..fin_n), bias in aux after bias_1; the residual stream itself is not written back
  float* fin_out;
  int64_t ld_fin;
  int fin_n;
  const float* xin;       // kernels_chain4.h CHAIN_IN: the noisy input [B][xin_C][rows_per_seq] fp32 (model/diffusion.py:345-346: permuted and projected inside the kernel)
  int xin_C;
  int fin_x3;             // kernels_chain4.h, last layer, <= 64 rows: final_layer as a split-operand exact island (stream: [W_hi | W_hi | W_lo]) into fin_out [m][ld_fin]; bias in aux after bias_1
  // diagnostic (A2P_CHAIN_CLK=1): blocks 0..7 write {s_memtime, s_memrealtime} at kernel begin / end to clk[block][2][2]: the
  // shader clock the kernel actually ran at inside the step (DVFS) = d(memtime) / d(memrealtime @ 100 MHz)
  unsigned long long* clk;
};

// one 16 KiB stage of a packed stream: stage = rows [row0, row0+128) x k [k0, k0+64) of W[., ldw]
struct ChainPackDesc {
  const h16_t* W;
  int ldw, row0, k0, nrows;  // rows >= nrows are zero-filled
  int omap;                  // 1: stage of a GEMM whose output tiles go straight to HBM (chain_body::gemm_store): the 8-wave slices use
                             // the PAIRED column map below (the 4-wave slices are the same for both values)
};

__global__ __launch_bounds__(256) void chain_pack_kernel(const ChainPackDesc* __restrict__ descs, h16_t* __restrict__ dst, int nw) {
  const ChainPackDesc d = descs[blockIdx.x];
  uint4* out = reinterpret_cast<uint4*>(dst + (int64_t)blockIdx.x * CHAIN_STAGE_ELEMS);
  const int cw = 128 / nw;  // out-cols (weight rows) per wave: 32 (4 waves) or 16 (8 waves)
  for (int q = threadIdx.x; q < 1024; q += 256) {  // 16-byte chunk q = (wave, row-in-slice, chunk position)
    const int w = q / (cw * 8), r = (q % (cw * 8)) >> 3, pos = q & 7;
    // Column ownership (chain_kernel::col_of): MFMA sub-tile row i = 4*g + e of 16-column sub-tile J of 32-column group W4 is
    // output column W4*32 + g*8 + J*4 + e, so that the two sub-tiles of a 4-wave lane are ADJACENT (8 contiguous columns per
    // lane: 16-byte bf16 stores / LDS writes, 32 contiguous bytes of the fp32 residual).  4 waves: W4 = w, J = r >> 4;
    // 8 waves: W4 = w >> 1, J = w & 1 -- the same column sets per (W4, J), which keeps the two shapes bit-identical.
    const int w4 = nw == 4 ? w : (w >> 1), J = nw == 4 ? (r >> 4) : (w & 1), i = r & 15;
    // Paired output map (8 waves, omap): a wave owns 16 columns of a tile = 32 bytes of an output row, and stores of less than
    // 64 contiguous bytes per row cost ~9 cycles per touched line against ~3.5 (scratch/issue_probe: 743 vs 392 cycles per stage).
    // So the wave's columns of tiles 2k and 2k+1 are made ADJACENT: column = 256*k + 64*W4 + 32*J + 16*(tile & 1) + i, and the
    // pair of tiles is written together, 64 contiguous bytes per row (chain_body::gemm_store).
    const int tile = d.row0 >> 7;
    const int row = (nw == 8 && d.omap) ? (tile >> 1) * 256 + w4 * 64 + J * 32 + (tile & 1) * 16 + i
                                        : d.row0 + w4 * 32 + (i >> 2) * 8 + J * 4 + (i & 3);
    const int chunk = pos ^ ((r >> 1) & 7);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < d.nrows) v = *reinterpret_cast<const uint4*>(d.W + (int64_t)row * d.ldw + d.k0 + chunk * 8);
    out[q] = v;
  }
}

__device__ __forceinline__ void chain_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Activation accesses go through these helpers so their cache policy is one switch.  Non-temporal (-DCHAIN_NT_ACT) was
// measured and rejected: 2.42 vs 2.37 ms per step at B=8 (same box, 3 alternating runs) -- the activations are re-read by
// the next kernel from L2/MALL, and evict-first costs more there than it saves on the weight stream.
#ifdef CHAIN_NT_ACT
#define CHAIN_NT 1
#else
#define CHAIN_NT 0
#endif
#ifdef CHAIN_NT_STORES   // experiment: only the output stores non-temporal (write-through: no dirty-line flush at kernel end)
#define CHAIN_NT_ST 1
#else
#define CHAIN_NT_ST CHAIN_NT
#endif
__device__ __forceinline__ f32x4 chain_ld4(const float* p) {
  if constexpr (CHAIN_NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  else return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ void chain_st4(float* p, f32x4 v) {
  if constexpr (CHAIN_NT_ST) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void chain_st_bf8(h16_t* p, h16x8 v) {
  if constexpr (CHAIN_NT_ST) __builtin_nontemporal_store(v, reinterpret_cast<h16x8*>(p));
  else *reinterpret_cast<h16x8*>(p) = v;
}
__device__ __forceinline__ void chain_st_bf4(h16_t* p, h16x4 v) {
  if constexpr (CHAIN_NT_ST) __builtin_nontemporal_store(v, reinterpret_cast<h16x4*>(p));
  else *reinterpret_cast<h16x4*>(p) = v;
}

// LDS-only barrier: never waits for the DMA queue
__device__ __forceinline__ void chain_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ABL: ablation switches for scratch/chain_bench.hip only (the library instantiates ABL = 0):
//   1 = no global stores, 2 = no MFMA, 4 = no weight DMA / waits, 8 = no workgroup barriers in the FFN,
//   16 = no fragment reads from LDS, 64 = phase time stamps (100 MHz s_memrealtime) of blocks 0 / 101 into p.fin_out,
//   debug (scratch/chain_dbg.hip): 256 = stop after the out_proj epilogue, 512 = out_proj result discarded, 1024 = FFN result discarded
// NW: waves per workgroup.  4 = one 512-register wave per SIMD; 8 = two 256-register waves per SIMD, each owning 16 of a
// tile's 128 columns (half the accumulators, half the weight slice, its own DMA ring): a wave's LDS-DMA pieces and
// fragment reads cost it 40-60 issue cycles each that its own MFMAs do not hide (measured additive, docs/lab_notebook_r1_r4.md section 4),
// so the second wave on the SIMD is what overlaps them.
// LDS footprint (bf16 elements) of one workgroup: [panelA BM x D][panelH BM x 128][LayerNorm partials][aux][weight ring]
template <int D, int MT>
struct ChainLds {
  static constexpr int BM = 16 * MT, AUX_F = 2560;
  static constexpr int FIXED = BM * D + BM * 128 + 32 * BM + 2 * AUX_F;
  static constexpr int NS = (160 * 1024 / 2 - FIXED) / CHAIN_STAGE_ELEMS > 6 ? 6 : (160 * 1024 / 2 - FIXED) / CHAIN_STAGE_ELEMS;
  static constexpr int ELEMS = FIXED + NS * CHAIN_STAGE_ELEMS;
};

// The chain of one row panel: rows [m0, m0 + 16*MT) of the launch, LDS at `smem` (ChainLds<D, MT>::ELEMS elements).
template <int D, int MT, int MODE, int ABL, int NW>
__device__ __forceinline__ void chain_body(const ChainP& p, h16_t* const smem, const int m0) {
  constexpr int CW = 128 / NW;   // columns of a 128-column tile owned by one wave
  constexpr int NJ = CW / 16;    // 16-column sub-tiles per wave per tile
  constexpr int PCS = CW / 8;    // 1 KiB LDS-DMA pieces per wave per stage
  constexpr int BM = 16 * MT;
  constexpr int CPR = D / 8;     // 16-byte chunks per panel row
  constexpr int NT = D / 128;    // 128-column tiles of a D-wide output
  constexpr int KS = D / 64;     // k-steps of a D-deep contraction
  constexpr int NSUB = NJ * NT;  // 16-column sub-tiles a wave owns of a D-wide output
  constexpr int FT = 8;          // ff_size / 128
  constexpr int HLD = 128;       // hidden chunk row stride
  constexpr int AUX_F = 2560;    // floats of per-tile biases (10 KiB)
  constexpr int NS = ChainLds<D, MT>::NS;
  static_assert(ChainLds<D, MT>::AUX_F == AUX_F && ChainLds<D, MT>::BM == BM && HLD == 128, "ChainLds out of step");
  static_assert(NS >= 3, "panel too tall for a 3-deep weight ring");
  constexpr int WSLICE = CHAIN_STAGE_ELEMS / NW;  // elements per wave per stage
  h16_t* const panelA = smem;
  h16_t* const panelH = panelA + BM * D;
  float* const red = reinterpret_cast<float*>(panelH + BM * HLD);  // [2][8][BM]: LayerNorm partial sums per 16-column group
  float* const aux = red + 16 * BM;                                  // [AUX_F]
  h16_t* const ring = reinterpret_cast<h16_t*>(aux + AUX_F);      // [wave][NS][32][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int W4 = NW == 4 ? wid : (wid >> 1), J0 = NW == 4 ? 0 : (wid & 1);  // 32-column group and first 16-column sub-tile of this wave
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if constexpr (ABL & 64) {
      if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101))
        reinterpret_cast<unsigned long long*>(p.fin_out)[(blockIdx.x ? 32 : 0) + i] = wall_clock64();
    }
  };
  stamp(0);
  if (p.clk && tid == 0 && blockIdx.x < 8) {
    p.clk[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
    p.clk[blockIdx.x * 4 + 1] = wall_clock64();
  }
  h16_t* const myring = ring + wid * NS * WSLICE;

  // ---- weight stream ---------------------------------------------------------------------------------------
  const h16_t* wsrc = p.stream + wid * WSLICE + lane * 8;  // this lane's 16 bytes of instruction 0 of stage 0
  // ring positions as running element offsets with a compare-and-wrap (a `% NS` with NS = 5 costs a multiply-high chain of
  // scalar instructions per stage, and with one wave per SIMD every instruction is 4 issue cycles)
  int issue_off = 0, consume_off = 0;
  auto issue_stage = [&]() __attribute__((always_inline)) {
    h16_t* buf = myring + issue_off;
    if constexpr (!(ABL & 4)) {  // 4 x 1 KiB; the instruction offset advances the global AND the LDS address (one M0 write per stage)
      const auto gp = (const __attribute__((address_space(1))) void*)wsrc;
      const auto lp = (__attribute__((address_space(3))) void*)buf;
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
      if constexpr (PCS == 4) {
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
    }
    wsrc += CHAIN_STAGE_ELEMS;  // past the end of the stream: the host pads CHAIN_STREAM_PAD stages
    issue_off = issue_off + WSLICE == NS * WSLICE ? 0 : issue_off + WSLICE;
  };
  // wait for this wave's oldest slice (NS-2 newer ones stay in flight), hand the just-freed slot to the DMA
  auto stage_begin = [&]() __attribute__((always_inline)) -> const h16_t* {
    // lgkmcnt(0): the fragment reads of the slot that is about to be refilled have returned
    if constexpr (!(ABL & 4)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PCS * (NS - 2)) : "memory");
    // pin the step boundary: hipcc otherwise hoists the NEXT step's MFMAs above this wait, right behind their fragment
    // reads, which un-pipelines the loop (cdna_hip_programming.md §5.4 rule 18)
    __builtin_amdgcn_sched_barrier(0);
    issue_stage();
    const h16_t* wb = myring + consume_off;
    consume_off = consume_off + WSLICE == NS * WSLICE ? 0 : consume_off + WSLICE;
    return wb;
  };
  // one k-step (64) of a [BM x 128] tile = this wave's fragments of 2 MFMA k-chunks.  Reads and MFMAs are split so the
  // reads of step s+1 are in flight while the MFMAs of step s issue (one wave per SIMD: nobody else hides LDS latency).
  struct Frags {
    h16x8 a[2][MT], w[2][NJ];
  };
  auto load_frags = [&](Frags& f, const h16_t* P, int pld, int kchunk0, const h16_t* wb) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if constexpr (ABL & 16) asm volatile("" : "=v"(f.a[kk][mt]));
        else f.a[kk][mt] = *reinterpret_cast<const h16x8*>(P + (mt * 16 + l15) * pld + (((kchunk0 + kk * 4 + g) ^ l15) << 3));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int wrow = j * 16 + l15;
        if constexpr (ABL & 16) asm volatile("" : "=v"(f.w[kk][j]));
        else f.w[kk][j] = *reinterpret_cast<const h16x8*>(wb + wrow * 64 + (((kk * 4 + g) ^ ((wrow >> 1) & 7)) << 3));
      }
    }
  };
  // swap = false: D = C^T, lane holds 4 consecutive columns n of row m = l15 (row-major consumers);
  // swap = true : D = C,   lane holds 4 consecutive rows m = g*4 + r of column n = l15 (the transposed V^T store)
  auto mma_frags = [&](f32x4(&acc)[MT][NJ], const Frags& f, bool swap) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (!(ABL & 2)) {
            if (swap) acc[mt][j] = A2P_MFMA16(f.a[kk][mt], f.w[kk][j], acc[mt][j]);
            else acc[mt][j] = A2P_MFMA16(f.w[kk][j], f.a[kk][mt], acc[mt][j]);
          } else {
            asm volatile("" ::"v"(f.w[kk][j]), "v"(f.a[kk][mt]));
          }
        }
  };
  // acc += P[:, 0:64*nks] * (the next nks stream stages)^T, nks even
  auto gemm_tile = [&](f32x4(&acc)[MT][NJ], const h16_t* P, int pld, int nks, bool swap = false) __attribute__((always_inline)) {
    // Issue order inside one step (one wave per SIMD, in-order issue): each MFMA occupies the matrix pipe for 16 cycles
    // but its issue slot for 4, so the DMA pieces and fragment reads of the NEXT step are slotted between the MFMAs
    // of the current one instead of in front of them.
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < PCS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (LDS-DMA piece)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      }
#pragma unroll
      for (int i = 0; i < 2 * MT + 2 * NJ; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      }
    };
    Frags f0, f1;
    load_frags(f0, P, pld, 0, stage_begin());
    for (int ks = 0; ks < nks; ks += 2) {
      load_frags(f1, P, pld, (ks + 1) * 8, stage_begin());
      mma_frags(acc, f0, swap);
      interleave();
      if (ks + 2 < nks) {
        load_frags(f0, P, pld, (ks + 2) * 8, stage_begin());
        mma_frags(acc, f1, swap);
        interleave();
      } else {
        mma_frags(acc, f1, swap);
      }
    }
  };
  // acc[t] += P[:, 0:64*NKS] * (stream stages)^T for all NT tiles of a group, as ONE pipelined pass over NKS*NT stream stages in
  // k-major order (stage = ks*NT + t; the host packs them so): the A fragments of a k-step are read once for all NT tiles instead of
  // once per tile (8 waves are LDS-read bound: every wave reads every panel row), and the group pays one pipeline ramp instead of
  // NT.  Per tile the k-order is unchanged: same bits as the tile-major form.
  auto gemm_group = [&](f32x4(&acc)[NT][MT][NJ], const h16_t* P, int pld, auto nks_c, bool swap = false) __attribute__((always_inline)) {
    constexpr int NKS = decltype(nks_c)::value, NST = NKS * NT;
    h16x8 a[2][2][MT], w[2][2][NJ];   // [buffer][k-chunk][...]
    auto load_a = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[buf][kk][mt] = *reinterpret_cast<const h16x8*>(P + (mt * 16 + l15) * pld + (((ks * 8 + kk * 4 + g) ^ l15) << 3));
    };
    auto load_w = [&](int buf, const h16_t* wb) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int wrow = j * 16 + l15;
          w[buf][kk][j] = *reinterpret_cast<const h16x8*>(wb + wrow * 64 + (((kk * 4 + g) ^ ((wrow >> 1) & 7)) << 3));
        }
    };
    load_a(0, 0);
    load_w(0, stage_begin());
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const int ks = s / NT, t = s % NT;
      const bool new_k = (s + 1) % NT == 0;   // the next stage starts a k-step
      if (s + 1 < NST) {
        load_w((s + 1) & 1, stage_begin());
        if (new_k) load_a((ks + 1) & 1, ks + 1);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (swap) acc[t][mt][j] = A2P_MFMA16(a[ks & 1][kk][mt], w[s & 1][kk][j], acc[t][mt][j]);
            else acc[t][mt][j] = A2P_MFMA16(w[s & 1][kk][j], a[ks & 1][kk][mt], acc[t][mt][j]);
          }
      if (s + 1 < NST) {
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 2 * NJ + (new_k ? 2 * MT : 0); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    }
  };
  auto zero = [&](f32x4(&acc)[MT][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // Per-tile bias reads from the LDS aux block go through inline asm: a compiler-visible ds_read at a runtime offset makes
  // hipcc emit s_waitcnt vmcnt(0) first (it cannot prove the read does not alias a pending LDS-DMA slice), which drained
  // the weight ring once per tile -- ~1 us x 20 tiles per POST kernel.  The aux block was written before the kernel's
  // first barrier, so no DMA ever targets it again.
  auto lds_off = [&](const float* q) __attribute__((always_inline)) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q;
  };
  // accumulators start at the per-column bias held in the LDS aux block
  auto init_bias = [&](f32x4(&acc)[MT][NJ], const float* bias_lds) __attribute__((always_inline)) {
    f32x4 b[2];
    if constexpr (NJ == 2)
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(b[0]), "=&v"(b[1])
                   : "v"(lds_off(bias_lds + W4 * 32 + g * 8))
                   : "memory");
    else
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b[0]) : "v"(lds_off(bias_lds + W4 * 32 + g * 8 + J0 * 4)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][j] = b[j];
  };
  // 8 waves, GEMMs that store their tiles (gemm_store): first of the 16 columns this wave owns of output tile t under the paired
  // column map of chain_pack_kernel -- tiles 2k and 2k+1 are adjacent halves of one 64-byte run per row
  auto obase = [&](int t) __attribute__((always_inline)) { return (t >> 1) * 256 + W4 * 64 + J0 * 32 + (t & 1) * 16; };
  auto init_bias_o = [&](f32x4(&acc)[MT][NJ], const float* bias_lds, int t) __attribute__((always_inline)) {
    f32x4 b;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b) : "v"(lds_off(bias_lds + obase(t) + g * 4)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = b;
  };
  auto init_bias_ot = [&](f32x4(&acc)[MT][NJ], const float* bias_lds, int t) __attribute__((always_inline)) {
    float b;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b) : "v"(lds_off(bias_lds + obase(t) + l15)) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = f32x4{b, b, b, b};
  };
  // same for the swapped (D = C) orientation: one bias value per lane column n = j*16 + l15
  auto init_bias_t = [&](f32x4(&acc)[MT][NJ], const float* bias_lds) __attribute__((always_inline)) {
    float b[2];
    if constexpr (NJ == 2)
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(b[0]), "=&v"(b[1])
                   : "v"(lds_off(bias_lds + W4 * 32 + (l15 >> 2) * 8 + (l15 & 3)))
                   : "memory");
    else
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b[0]) : "v"(lds_off(bias_lds + W4 * 32 + (l15 >> 2) * 8 + J0 * 4 + (l15 & 3))) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][j] = f32x4{b[j], b[j], b[j], b[j]};
  };
  // first of the 4 consecutive output columns this lane holds of sub-tile (tile t, half j): see chain_pack_kernel.  With 4 waves
  // the two halves of a lane are adjacent (col_of(t, 1) == col_of(t, 0) + 4)
  auto col_of = [&](int t, int j) __attribute__((always_inline)) { return t * 128 + W4 * 32 + g * 8 + (J0 + j) * 4; };

  // float offset of this lane's 4 columns of sub-tile (t, j) of row m in the tiled residual layout (ChainP::x_in_tiled)
  auto x_tiled_off = [&](int m, int t, int j) __attribute__((always_inline)) {
    return ((int64_t)(m >> 4) * (D / 16) + (t * 8 + W4 * 2 + J0 + j)) * 256 + (g * 16 + (m & 15)) * 4;
  };
  // ---- kernel start: panel + aux DMA, residual rows, stream prefetch ----------------------------------------
  f32x4 xrow[MT][NSUB];
  int row_m[MT], row_seq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + mt * 16 + l15;
    m = m < p.M ? m : p.M - 1;
    row_m[mt] = m;
    row_seq[mt] = m / p.rows_per_seq;
  }
  if constexpr (MODE != CHAIN_PRE) {  // attention output panel: one global_load_lds per 64 chunks
    constexpr int RPI = 64 / CPR;     // rows per wave instruction (1 for d = 512, 2 for d = 256)
    for (int r0 = wid * RPI; r0 < BM; r0 += NW * RPI) {
      const int row = r0 + lane / CPR, pos = lane % CPR;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      m = (p.src_rows > 0 && m >= p.src_rows) ? m - p.src_rows : m;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ain + (int64_t)m * p.ld_ain + ((pos ^ (row & 15)) << 3)),
                                       (__attribute__((address_space(3))) void*)(panelA + r0 * D), 16, 0, CHAIN_NT ? 2 : 0);
    }
  }
  for (int kb = wid; kb < p.aux_kb; kb += NW) chain_glds16(p.aux + kb * 256 + lane * 4, aux + kb * 256);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int ms = (p.src_rows > 0 && row_m[mt] >= p.src_rows) ? row_m[mt] - p.src_rows : row_m[mt];
      xrow[mt][ns] = chain_ld4((p.xsrc ? p.xsrc : p.x) + (p.x_in_tiled ? x_tiled_off(ms, ns / NJ, ns % NJ) : (int64_t)ms * D + col_of(ns / NJ, ns % NJ)));
    }
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue_stage();
  // everything older than the NS-1 weight slices (panel, aux, residual rows) has landed for this wave ...
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS * (NS - 1)) : "memory");
  chain_bar();  // ... and for every other wave
  stamp(1);

  // ---- epilogue helpers ---------------------------------------------------------------------------------------
  // FiLM affine + residual (transformer_modules.py:122-124,193): x += (scale + 1) * (acc + bias) + shift.
  // All operands of a tile are fetched in one batch BEFORE the arithmetic, and `film != NULL` is tested once per tile:
  // a per-element "if (film) load" made hipcc branch around every load and wait for each one separately
  // (24 dependent L2 round trips per tile, cdna_hip_programming.md §5 "three .s-level traps" (c)).
  auto film_res = [&](f32x4(&acc)[MT][NJ], int t, const float* bias, const float* film) __attribute__((always_inline)) {
    f32x4 b[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const f32x4*>(bias + col_of(t, j));
    if (film) {
      // tall panels (MT >= 4 with 512-register waves): one batch per column half instead of one per tile -- half the live FiLM
      // operands (the single-batch form spilled into the FFN loop), one more L2 round trip per tile
      constexpr int JB = (MT >= 4 && NW == 4) ? 1 : NJ;
#pragma unroll
      for (int j0 = 0; j0 < NJ; j0 += JB) {
        f32x4 sc[JB][MT], sh[JB][MT];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float* fp = film + (int64_t)row_seq[mt] * p.film_seq_stride + col_of(t, j0 + j);
            sc[j][mt] = *reinterpret_cast<const f32x4*>(fp);
            sh[j][mt] = *reinterpret_cast<const f32x4*>(fp + p.film_shift_off);
          }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          asm volatile("" : "+v"(b[j0 + j]));
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(sc[j][mt]), "+v"(sh[j][mt]));
        }
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4& xr = xrow[mt][t * NJ + j0 + j];
            const f32x4 y = acc[mt][j0 + j] + b[j0 + j], s1 = sc[j][mt] + 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[e] += fmaf(s1[e], y[e], sh[j][mt][e]);
          }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xrow[mt][t * NJ + j] += acc[mt][j] + b[j];
    }
  };
  // LayerNorm statistics of the register rows (eps 1e-5, biased variance, two-pass like ln_rope_kernel)
  float ln_mean[MT], ln_rstd[MT];
  // The row statistics are reduced over EIGHT partials per row -- one per 16-column group of a 128-column tile, i.e. per wave
  // with NW = 8 and per (wave, half) with NW = 4 -- in one fixed tree, so both workgroup shapes produce bit-identical rows
  // (the run-time choice between them, a2p_lib_run.h `chain_pick_nw`, is then invisible in the results).
  auto group_partials = [&](const float* q) __attribute__((always_inline)) {
    return ((q[0] + q[BM]) + (q[2 * BM] + q[3 * BM])) + ((q[4 * BM] + q[5 * BM]) + (q[6 * BM] + q[7 * BM]));
  };
  auto ln_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float s[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) s[j] = 0.f;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) s[ns % NJ] += (xrow[mt][ns][0] + xrow[mt][ns][1]) + (xrow[mt][ns][2] + xrow[mt][ns][3]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = s[j];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) red[(wid * NJ + j) * BM + mt * 16 + l15] = v;
      }
    }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = mt * 16 + l15;
      ln_mean[mt] = group_partials(red + r) * (1.0f / D);
      float q[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) q[j] = 0.f;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = xrow[mt][ns][e] - ln_mean[mt];
          q[ns % NJ] = fmaf(dlt, dlt, q[ns % NJ]);
        }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = q[j];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) red[8 * BM + (wid * NJ + j) * BM + r] = v;
      }
    }
    chain_bar();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = mt * 16 + l15;
      const float var = group_partials(red + 8 * BM + r) * (1.0f / D);
      ln_rstd[mt] = 1.0f / sqrtf(var + 1e-5f);
    }
  };
  // normalised (optionally rotated, rotary_embedding_torch.py:46-66) rows -> bf16 A panel.
  // All global operands (gamma, beta, the rows' cos/sin entries) are fetched in ONE batch and pinned before the arithmetic:
  // left alone, hipcc sinks each load next to its use and waits for it there -- 24+ dependent L2 round trips per call.
  auto ln_write = [&](const float* gamma, const float* beta, auto rope_c) __attribute__((always_inline)) {
    constexpr bool ROPE = decltype(rope_c)::value;
    f32x4 ga[NSUB], be[NSUB];
    f32x4 cs[ROPE ? MT : 1][ROPE ? NSUB : 1];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int n = col_of(ns / NJ, ns % NJ);
      ga[ns] = *reinterpret_cast<const f32x4*>(gamma + n);
      be[ns] = *reinterpret_cast<const f32x4*>(beta + n);
      if constexpr (ROPE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int pos = row_m[mt] - row_seq[mt] * p.rows_per_seq;
          cs[mt][ns] = p.cst[(int64_t)(n >> 2) * p.cs_npos + pos];  // (cos,sin) x 2; the 16 rows of a lane group are 256 contiguous bytes
        }
      }
    }
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      asm volatile("" : "+v"(ga[ns]), "+v"(be[ns]));
      if constexpr (ROPE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(cs[mt][ns]));
      }
    }
    // rows -> bf16 panel: with 4 waves a lane's two sub-tiles are adjacent columns, so one 16-byte LDS write per (row, tile)
    auto norm4 = [&](int mt, int ns) __attribute__((always_inline)) -> h16x4 {
      // (x - mean) * rstd as one fma per element: x * rstd + (-mean * rstd)
      const float rs = ln_rstd[mt], nm = -ln_mean[mt] * rs;
      float v0 = fmaf(fmaf(xrow[mt][ns][0], rs, nm), ga[ns][0], be[ns][0]);
      float v1 = fmaf(fmaf(xrow[mt][ns][1], rs, nm), ga[ns][1], be[ns][1]);
      float v2 = fmaf(fmaf(xrow[mt][ns][2], rs, nm), ga[ns][2], be[ns][2]);
      float v3 = fmaf(fmaf(xrow[mt][ns][3], rs, nm), ga[ns][3], be[ns][3]);
      if constexpr (ROPE) {
        const f32x4 t = cs[mt][ns];
        const float r0 = fmaf(v0, t[0], -(v1 * t[1])), r1 = fmaf(v1, t[0], v0 * t[1]);
        const float r2 = fmaf(v2, t[2], -(v3 * t[3])), r3 = fmaf(v3, t[2], v2 * t[3]);
        v0 = r0; v1 = r1; v2 = r2; v3 = r3;
      }
      return h16x4{(h16_t)v0, (h16_t)v1, (h16_t)v2, (h16_t)v3};
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = col_of(t, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        h16_t* dst = panelA + (mt * 16 + l15) * D + ((((n >> 3) ^ l15) << 3) | (n & 7));
        if constexpr (NJ == 2) {
          const h16x4 lo = norm4(mt, t * NJ), hi = norm4(mt, t * NJ + 1);
          *reinterpret_cast<h16x8*>(dst) = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          *reinterpret_cast<h16x4*>(dst) = norm4(mt, t);
        }
      }
    }
    chain_bar();  // the panel is complete before any wave's fragment reads
  };
  auto store_x = [&]() __attribute__((always_inline)) {
    if constexpr (ABL & 1) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (m0 + mt * 16 + l15 >= p.M) continue;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns)
        chain_st4(p.x + (p.x_out_tiled ? x_tiled_off(row_m[mt], ns / NJ, ns % NJ) : (int64_t)row_m[mt] * D + col_of(ns / NJ, ns % NJ)), xrow[mt][ns]);
    }
  };
  // D-deep GEMM over `ntiles` output tiles with a per-tile bf16 store: out[m][n] (row-major, 4 columns per lane) or the
  // transposed V^T layout out[seq][n][t] (operands swapped: 4 consecutive frames t per lane, one 8-byte store each)
  // `fin`: this is the kernel's last GEMM.  Before the stores of its last tile the weight DMA is drained (the ring runs up to
  // NS-1 stages past the end of the stream and must have landed before the LDS is released), so that the kernel can end with
  // its final stores -- the last tile's and the residual rows' -- still in flight instead of waiting for their acknowledgement.
  [[maybe_unused]] int gs_stamp = 13;   // diagnostic build: next free stamp slot for gemm_store's per-tile stamps
  auto gemm_store = [&](int ntiles, const float* bias_lds, h16_t* out, int64_t ldo, bool transposed, bool fin = false) __attribute__((always_inline)) {
    // V^T through LDS (frame count a multiple of 8): the accumulators hold 4 consecutive frames of one column per lane, i.e. an
    // 8-byte store per lane with 64 different 8-byte segments per instruction -- measured 150 issue cycles per instruction
    // against 45-60 for 16-byte stores (scratch/issue_probe "V^T pattern": +1.6 us per tile).  Each wave therefore transposes ITS
    // [CW columns][BM frames] block through a private slice of the (idle) hidden-chunk buffer and writes 16-byte pieces = 8
    // consecutive frames of one column, 2*MT adjacent pieces per column.  Same bytes, a quarter of the store instructions' cost.
    constexpr int VP = (CW * BM / 8 + 63) / 64;   // 16-byte pieces per lane
    const bool vt_staged = transposed && (p.rows_per_seq & 7) == 0;
    h16_t* const stg = panelH + wid * (CW * BM);
    int64_t voff[VP];
    int vcol[VP];
    if (vt_staged) {
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int q = lane + 64 * i, c = q / (BM / 8), m = m0 + (q % (BM / 8)) * 8;
        vcol[i] = NW == 8 ? c : W4 * 32 + ((c & 15) >> 2) * 8 + (J0 + (c >> 4)) * 4 + (c & 3);   // 8 waves: + obase(t), paired map
        const int sq = m / p.rows_per_seq;
        voff[i] = (q < CW * BM / 8 && m < p.M) ? (int64_t)sq * p.vt_seq_stride + (m - sq * p.rows_per_seq) : -1;
      }
    }
    [[maybe_unused]] h16x4 held[MT];   // 8 waves: the even tile of a pair, kept until its neighbour is done
    for (int t = 0; t < ntiles; ++t) {
      if constexpr (ABL & 64) { if (t > 0 && gs_stamp < 31) stamp(gs_stamp++); }   // previous tile's epilogue done
      f32x4 acc[MT][NJ];
      if constexpr (NW == 8) {
        if (!transposed) init_bias_o(acc, bias_lds, t);
        else init_bias_ot(acc, bias_lds, t);
      } else {
        if (!transposed) init_bias(acc, bias_lds + t * 128);
        else init_bias_t(acc, bias_lds + t * 128);
      }
      gemm_tile(acc, panelA, D, KS, transposed);
      if constexpr (ABL & 64) { if (gs_stamp < 31) stamp(gs_stamp++); }   // diagnostic build: tile GEMM done / tile stored (below)
      if (fin && t == ntiles - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (NW == 8) {
        if (!transposed) {
          // Tiles 2k | 2k+1 of this wave are the two 32-byte halves of one 64-byte run per row (paired map).  The pair goes
          // through the wave's slice of the idle hidden-chunk buffer, one 16-row tile at a time ([16 rows][32 columns] = 1 KiB),
          // and leaves as 16-byte pieces: lane q writes piece q & 3 of row q >> 2, i.e. 16 rows x 64 contiguous bytes per store.
          if ((t & 1) == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 v = acc[mt][0];
              held[mt] = h16x4{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            }
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 v = acc[mt][0];
              const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
              // [16 rows][32 columns] with the two 32-byte halves of rows 4..7 and 12..15 swapped: rows r, r+4, r+8, r+12 share
              // their LDS banks (64-byte pitch), and un-swizzled the 16 rows of one ds_write_b64 hit the same 4 slots 4 times
              // (scratch/lds_probe: 128 vs 32 cycles per instruction with 8 waves writing)
              const int hsw = (l15 >> 2) & 1;
              const uint32_t wa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(stg + l15 * 32 + hsw * 16 + g * 4);
              const uint32_t wb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(stg + l15 * 32 + (hsw ^ 1) * 16 + g * 4);
              asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" ::"v"(wa), "v"(wb), "v"(held[mt]), "v"(o) : "memory");
              h16x8 w;
              const int prow = lane >> 2, pp = lane & 3;   // this lane stores piece pp (8 columns) of row prow
              asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                           : "=v"(w)
                           : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(stg + prow * 32 + (((pp >> 1) ^ ((prow >> 2) & 1)) * 2 + (pp & 1)) * 8))
                           : "memory");
              if constexpr (ABL & 1) {
                asm volatile("" ::"v"(w));
                continue;
              }
              const int m = m0 + mt * 16 + (lane >> 2);
              if (m < p.M) chain_st_bf8(out + (int64_t)m * ldo + obase(t - 1) + (lane & 3) * 8, w);
            }
          }
          continue;
        }
      }
      if (!transposed) {  // 4 waves: 8 contiguous columns per lane, one 16-byte store per row
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (ABL & 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[mt][j]));
            continue;
          }
          if (m0 + mt * 16 + l15 >= p.M) continue;
          h16_t* dst = out + (int64_t)row_m[mt] * ldo + col_of(t, 0);
          const f32x4 v = acc[mt][0];
          if constexpr (NJ == 2) {
            const f32x4 u = acc[mt][1];
            chain_st_bf8(dst, h16x8{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3], (h16_t)u[0], (h16_t)u[1], (h16_t)u[2], (h16_t)u[3]});
          } else {
            chain_st_bf4(dst, h16x4{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]});
          }
        }
      } else if (vt_staged) {
        // inline-asm LDS accesses: a compiler-visible one would be ordered behind every pending LDS-DMA slice (vmcnt(0))
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[mt][j];
            const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(stg + (j * 16 + l15) * BM + mt * 16 + g * 4)), "v"(o) : "memory");
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < VP; ++i) {
          h16x8 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(stg + (lane + 64 * i) * 8)) : "memory");
          if constexpr (ABL & 1) {
            asm volatile("" ::"v"(v));
            continue;
          }
          if (voff[i] >= 0) chain_st_bf8(out + voff[i] + (int64_t)(NW == 8 ? obase(t) + vcol[i] : t * 128 + vcol[i]) * ldo, v);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[mt][j];
            if constexpr (ABL & 1) {
              asm volatile("" ::"v"(v));
              continue;
            }
            const h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            // rows m .. m+3 (m % 4 == 0) of column n: one 8-byte store when the frame count is a multiple of 4 (the four rows
            // then share a sequence and the address is aligned), else row by row (T = 30 k frames, k odd)
            const int m = m0 + mt * 16 + g * 4;
            if (m >= p.M) continue;
            const int sq = m / p.rows_per_seq, n = NW == 8 ? obase(t) + l15 : t * 128 + W4 * 32 + (l15 >> 2) * 8 + (J0 + j) * 4 + (l15 & 3);
            if ((p.rows_per_seq & 3) == 0) {
              chain_st_bf4(out + (int64_t)sq * p.vt_seq_stride + (int64_t)n * ldo + (m - sq * p.rows_per_seq), o);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int mm = m + r;
                if (mm >= p.M) break;
                const int s2 = mm / p.rows_per_seq;
                out[(int64_t)s2 * p.vt_seq_stride + (int64_t)n * ldo + (mm - s2 * p.rows_per_seq)] = o[r];
              }
            }
          }
        }
      }
    }
  };
  // norm1 -> rotary -> [Q|K] ; norm1 -> V^T          (aux: bias_qk at aq, bias_v right after)
  auto pre_work = [&](const float* aq) __attribute__((always_inline)) {   // always the kernel's last GEMMs
    ln_stats();
    ln_write(p.lnB_g, p.lnB_b, std::true_type{});
    stamp(9);
    gemm_store(2 * NT, aq, p.qk_out, p.ld_qk, false);
    chain_bar();  // every wave is done reading the rotated panel
    stamp(10);
    ln_write(p.lnB_g, p.lnB_b, std::false_type{});
    gemm_store(NT, aq + 2 * D, p.vt_out, p.ld_vt, true, true);
    stamp(11);
  };

  // ================================================================================================
  bool dma_drained = false;   // the path already waited for the run-ahead weight slices (gemm_store `fin`)
  if constexpr (MODE == CHAIN_PRE) {
    pre_work(aux);
    dma_drained = true;
  } else {
    // out_proj of the attention that produced `ain`; FiLM + residual into the register rows once all tiles are done
    auto out_proj_block = [&](const float* bias, const float* film) __attribute__((always_inline)) {
      f32x4 oacc[NT][MT][NJ];
#pragma unroll
      for (int t = 0; t < NT; ++t) zero(oacc[t]);
      gemm_group(oacc, panelA, D, std::integral_constant<int, KS>{});
      stamp(2);
      if constexpr (!(ABL & 512)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) film_res(oacc[t], t, bias, film);
      }
    };
    out_proj_block(p.bias_o, p.film_o);
    stamp(3);
    if constexpr (ABL & 256) {
      store_x();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    ln_stats();
    stamp(4);
    if constexpr (MODE == CHAIN_MID) {
      // the residual rows are written LAST: loads return in issue order behind stores, so the rotary-table loads of ln_write
      // (and every weight slice after them) issued behind 96 KB of row stores waited for their write acknowledgement
      ln_write(p.lnA_g, p.lnA_b, std::true_type{});
      gemm_store(NT, aux, p.q_out, p.ld_q, false, true);
      store_x();
      dma_drained = true;
    } else {
      if constexpr (MODE == CHAIN_MIDPOST) {
        // ---- the keyframe cross attention (multihead_attn2) without leaving the kernel ----
        static_assert(MODE != CHAIN_MIDPOST || D == 256, "the fused keyframe attention is written for the body model (8 heads x 32)");
        ln_write(p.lnA_g, p.lnA_b, std::true_type{});   // norm2a + rotary -> A panel
        {
          // query projection of ALL tiles into registers (k-major group GEMM, like out_proj), then over the panel it was computed from
          f32x4 qacc[NT][MT][NJ];
#pragma unroll
          for (int t = 0; t < NT; ++t) zero(qacc[t]);
          gemm_group(qacc, panelA, D, std::integral_constant<int, KS>{});
          chain_bar();   // every wave is done reading the rotated rows
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const int n = col_of(t, j);
              const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias_q2 + n);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                const f32x4 v = qacc[t][mt][j] + b;
                *reinterpret_cast<h16x4*>(panelA + (mt * 16 + l15) * D + ((((n >> 3) ^ l15) << 3) | (n & 7))) =
                    h16x4{(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
              }
            }
        }
        chain_bar();     // the query panel is complete
#ifndef A2P_KF_SKIP   // (scratch timing build: the kernel without its attention phase)
        {
          // S^T = K Q^T and O^T = V^T P^T per (16-row tile, head) with the fragment / key mapping of kernels_attn.h: lane (l15, g) owns
          // query row l15 and the 8 keys 8g .. 8g+7, so the probabilities feed the second MFMA without data movement.  Wave w takes
          // head(s) w * HPW ..; the keys of a head stay in 4 registers for all row tiles.  A tile that straddles two sequences is
          // computed against both caches and each row keeps its own.
          constexpr int DH = 32, HPW = (D / DH) / NW;
          const int nk = p.n_key2;
          auto slot_of = [&](int seq) __attribute__((always_inline)) {
            if (p.kv2_rule == 1) return 1 + seq;
            if (p.kv2_rule == 2) return 0;
            if (p.kv2_rule == 3) return seq < p.kv2_b ? 1 + seq : 0;
            return p.kv2_slots ? p.kv2_slots[seq] : seq;
          };
          float mtop = -INFINITY;
#pragma unroll
          for (int hh = 0; hh < HPW; ++hh) {
            const int h = wid * HPW + hh;
            h16x8 kf[2], vf[2];
            int cur = -1;
            auto load_kv = [&](int slot) __attribute__((always_inline)) {
              const h16_t* Kb = p.k2 + (int64_t)slot * p.k2_slot_stride + h * DH + g * 8;
              const h16_t* Vb = p.vt2 + (int64_t)slot * p.vt2_slot_stride + (int64_t)(h * DH + l15) * p.ld_vt2 + g * 8;
#pragma unroll
              for (int kt = 0; kt < 2; ++kt) {
                int key = 8 * (l15 >> 2) + 4 * kt + (l15 & 3);   // AttnLds::krow: accumulator rows 4g + r of both tiles = keys 8g .. 8g+7
                key = key < nk ? key : nk - 1;                   // absent keys: any valid row (their scores are masked)
                kf[kt] = *reinterpret_cast<const h16x8*>(Kb + (int64_t)key * p.ld_k2);
                vf[kt] = *reinterpret_cast<const h16x8*>(Vb + (int64_t)(kt * 16) * p.ld_vt2);   // (kt here: 16-row tile of head_dim)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (g * 8 + e >= nk) vf[kt][e] = (h16_t)0.f;   // never-written cache columns must not meet P = 0 as inf / nan
              }
            };
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int mf = m0 + mt * 16 < p.M ? m0 + mt * 16 : p.M - 1, ml = m0 + mt * 16 + 15 < p.M ? m0 + mt * 16 + 15 : p.M - 1;
              const int sA = slot_of(mf / p.rows_per_seq), sB = slot_of(ml / p.rows_per_seq);   // wave-uniform
              h16_t* const qp = panelA + (mt * 16 + l15) * D;
              const h16x8 qf = *reinterpret_cast<const h16x8*>(qp + (((h * 4 + g) ^ l15) << 3));
              auto attend = [&](f32x4(&o)[2]) __attribute__((always_inline)) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 sc[2] = {A2P_MFMA16(kf[0], qf, z), A2P_MFMA16(kf[1], qf, z)};
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    if (g * 8 + kt * 4 + r >= nk) sc[kt][r] = -INFINITY;
                    mx = fmaxf(mx, sc[kt][r]);
                  }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mtop = fmaxf(mtop, mx);
                const float sl = p.scale2 * 1.4426950408889634f, nm = -mx * sl;
                float l = 0.f;
                h16x8 pf;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], sl, nm));
                    l += e;
                    pf[kt * 4 + r] = (h16_t)e;
                  }
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
                const float inv = 1.0f / l;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                  o[dt] = A2P_MFMA16(vf[dt], pf, z);
#pragma unroll
                  for (int r = 0; r < 4; ++r) o[dt][r] *= inv;
                }
              };
              f32x4 oa[2];
              if (sA != cur) { load_kv(sA); cur = sA; }
              attend(oa);
              if (sB != sA) {   // the tile straddles two sequences
                f32x4 ob[2];
                load_kv(sB);
                cur = sB;
                attend(ob);
                const bool second = slot_of(row_seq[mt]) == sB;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) oa[dt][r] = second ? ob[dt][r] : oa[dt][r];
              }
#pragma unroll
              for (int dt = 0; dt < 2; ++dt) {
                const int n = h * DH + dt * 16 + 4 * g;
                *reinterpret_cast<h16x4*>(qp + ((((n >> 3) ^ l15) << 3) | (n & 7))) =
                    h16x4{(h16_t)oa[dt][0], (h16_t)oa[dt][1], (h16_t)oa[dt][2], (h16_t)oa[dt][3]};
              }
            }
          }
          if (p.stat_max) {   // largest row maximum of the scaled scores (a2p_attention_logit_max), guarded like kernels_attn.h
            float m = mtop * p.scale2;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const int bits = __float_as_int(m), mi = bits >= 0 ? bits : bits ^ 0x7fffffff;
            if (lane == 0 && mi > __atomic_load_n(p.stat_max, __ATOMIC_RELAXED)) atomicMax(p.stat_max, mi);
          }
        }
#endif
        chain_bar();     // the attention output panel is complete: the state a POST kernel starts from
        out_proj_block(p.bias_o2, p.film_o2);
        ln_stats();
      }
      ln_write(MODE == CHAIN_MIDPOST ? p.lnC_g : p.lnA_g, MODE == CHAIN_MIDPOST ? p.lnC_b : p.lnA_b, std::false_type{});
      stamp(5);
      // feed forward, split-K over the 8 hidden chunks: linear1 chunk -> GELU -> LDS -> linear2 partial
      f32x4 facc[NT][MT][NJ];
#pragma unroll
      for (int t = 0; t < NT; ++t) zero(facc[t]);
      for (int h = 0; h < FT; ++h) {
        f32x4 acc[MT][NJ];
        init_bias(acc, aux + h * 128);
        gemm_tile(acc, panelA, D, KS);
        if (h > 0 && !(ABL & 8)) chain_bar();  // every wave finished the linear2 partial of the previous chunk
        {
          const int c = W4 * 32 + g * 8 + J0 * 4;  // first column inside the hidden chunk (col_of within the tile)
          auto gelu4 = [&](const f32x4 v) __attribute__((always_inline)) {
            return h16x4{(h16_t)act_gelu_fast(v[0]), (h16_t)act_gelu_fast(v[1]), (h16_t)act_gelu_fast(v[2]), (h16_t)act_gelu_fast(v[3])};
          };
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            h16_t* dst = panelH + (mt * 16 + l15) * HLD + ((((c >> 3) ^ l15) << 3) | (c & 7));
            if constexpr (NJ == 2) {
              const h16x4 lo = gelu4(acc[mt][0]), hi = gelu4(acc[mt][1]);
              *reinterpret_cast<h16x8*>(dst) = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
              *reinterpret_cast<h16x4*>(dst) = gelu4(acc[mt][0]);
            }
          }
        }
        if (!(ABL & 8)) chain_bar();  // the hidden chunk is complete
        gemm_group(facc, panelH, HLD, std::integral_constant<int, 2>{});
      }
      stamp(6);
      if constexpr (!(ABL & 1024)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) film_res(facc[t], t, p.bias_2, p.film_f);
      }
      stamp(7);
      if (p.has_next == 2) {
        // final_layer on the finished rows: plain bf16 cast into the A panel, [BM x fin_n] GEMM, fp32 store
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          const int n = col_of(ns / NJ, ns % NJ);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const h16x4 o = {(h16_t)xrow[mt][ns][0], (h16_t)xrow[mt][ns][1], (h16_t)xrow[mt][ns][2], (h16_t)xrow[mt][ns][3]};
            *reinterpret_cast<h16x4*>(panelA + (mt * 16 + l15) * D + ((((n >> 3) ^ l15) << 3) | (n & 7))) = o;
          }
        }
        chain_bar();
        for (int t = 0; t < (p.fin_n + 127) / 128; ++t) {
          f32x4 acc[MT][NJ];
          init_bias(acc, aux + FT * 128 + t * 128);
          gemm_tile(acc, panelA, D, KS);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = col_of(t, j);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if (m0 + mt * 16 + l15 >= p.M || n >= p.fin_n) continue;
              if constexpr (!(ABL & 1))
                chain_st4(p.fin_out + (int64_t)row_m[mt] * p.ld_fin + n, acc[mt][j]);
            }
          }
        }
      } else {
        stamp(8);
        if (p.has_next) pre_work(aux + FT * 128);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_x();   // last, see MODE_MID
        dma_drained = true;
      }
    }
  }
  if (!dma_drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead DMA slices must land before the LDS is released
  if (p.clk && tid == 0 && blockIdx.x < 8) {
    p.clk[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();
    p.clk[blockIdx.x * 4 + 3] = wall_clock64();
  }
  stamp(12);
}

template <int D, int MT, int MODE, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void chain_kernel(const ChainP p) {
  __shared__ __attribute__((aligned(16))) h16_t smem[ChainLds<D, MT>::ELEMS];
  chain_body<D, MT, MODE, ABL, NW>(p, smem, blockIdx.x * (16 * MT));
}

// Two panel heights in ONE launch: workgroups [0, p.n_tall) take 16*MTA rows each, the rest 16*MTB (MTA > MTB).  A forward of
// more rows than 256 x 16*MTB runs in rounds over the 256 CUs (one workgroup per CU: the LDS) and the last round is mostly
// empty -- B=32, d=512: 800 panels of 48 rows = 3.125 rounds, i.e. FOUR rounds with 224 CUs idle in the last one.  With 96
// panels of 64 rows dispatched first and 672 of 48 rows behind them every CU gets exactly three panels (160 or 144 rows).
// Every row's result is independent of the panel height (tests/test_hip_round2.py), so the mix is invisible in the output.
template <int D, int MTA, int MTB, int MODE, int NW>
__global__ __launch_bounds__(64 * NW, 1) void chain_kernel_mix(const ChainP p) {
  __shared__ __attribute__((aligned(16))) h16_t smem[ChainLds<D, MTB>::ELEMS > ChainLds<D, MTA>::ELEMS ? ChainLds<D, MTB>::ELEMS : ChainLds<D, MTA>::ELEMS];
  const int b = blockIdx.x;   // wave-uniform
  if (b < p.n_tall) chain_body<D, MTA, MODE, 0, NW>(p, smem, b * (16 * MTA));
  else chain_body<D, MTB, MODE, 0, NW>(p, smem, p.n_tall * (16 * MTA) + (b - p.n_tall) * (16 * MTB));
}
#pragma clang fp contract(fast)
