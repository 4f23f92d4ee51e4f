// Fused output tail of the body model (round 4): final_layer (model/diffusion.py:397) + the six dilated convolutions of
// _run_single_pose_conv (:214-224: k = 3, dilations 1 2 3 1 2 3, 104 -> 256 -> 104 -> 104 ..., leaky ReLU 0.2, the
// (output[..., -len(y):] + y) / 2 skip connections, left padding of receptive_field - 1 = 24 zero frames) + final_conv (:401), as
// ONE kernel whose activations never leave the LDS.  Rounds 2-3 ran this as 9 split-operand GEMM launches of 23-26 us each
// (12 % of the body step for 0.6 % of its FLOPs: every launch is one HBM round trip of a [19200 x 128|256] activation).
//
// A workgroup (8 waves: 2 row halves x 4 column groups) owns 104 output frames of one sequence.  The convolutions are causal in
// the reference's right-aligned indexing -- y[f] = sum_k W[:, :, k] . in[f - (2 - k) * dilation] -- so the block recomputes a halo
// of 24 frames on its left (128 rows = eight 16-row MFMA tiles) from the residual stream and no workgroup ever waits for another.
// Frames before the start of the sequence enter as zeros behind final_layer (F.pad), exactly like the reference; deeper layers are
// computed from them, not zeroed.
// (The first version took 64 frames + 32 halo rows with a 256-channel LDS buffer: 320 workgroups at the bench shape = two rounds of
// the 256 CUs, 95 us.  128-row blocks need both LDS buffers at 128 channels -- the 256-channel operands pass through in two halves
// under accumulators that stay in registers -- and give 192 workgroups: one round.)
//
// Arithmetic: the exact island of the 16-bit modes (docs/lab_notebook_r1_r4.md section 4.3b) -- every operand is a (hi, lo) pair of 16-bit values
// and a . w = a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32 accumulators (the dropped a_lo w_lo is 2^-22 relative with IEEE half).
// Activations live in LDS as [row][hi: CP | lo: CP] 16-bit rows, 16-byte chunks XOR-swizzled by (row & 15) (conflict-free
// ds_read_b128 of 16 consecutive rows, also when a tap shifts the rows); weights stream from L2 as ready-made MFMA operands
// ([layer][k-chunk][16-column tile][hi | lo][64 lanes][8]: one global_load_dwordx4 of a wave = one operand), packed once at
// a2p_finalize_weights (tail_pack_kernel).
//
// Work per workgroup: 20.8 k MFMA 16x16x32 (three per product) + 3.4 MB of weight operands through L2; 192 workgroups at the
// bench shape (32 sequences x 600 frames = 6 blocks each).
#pragma once
#include "a2p_common.h"

#pragma clang fp contract(off)

namespace tail {
constexpr int HALO = 24;          // rows computed in front of the output frames: the receptive field - 1 of the six convolutions
constexpr int ROWS = 128;         // rows per workgroup = 2 halves x 4 row tiles of 16
constexpr int TB = ROWS - HALO;   // 104 output frames per workgroup (T = 600: 6 blocks per sequence, 192 workgroups = one round of the CUs)
constexpr int GUARD = 8;          // zero rows in front of the computed ones: taps of the first (never used) rows stay inside the buffer
constexpr int LROWS = ROWS + GUARD;
constexpr int CL = 128;           // channels per LDS row of BOTH buffers (104 padded to 128; 256-channel operands pass through in two halves)
constexpr int BUF = LROWS * 4 * CL;   // bytes per buffer: [row][hi CL | lo CL] 16-bit  (2 x 69 632 B of the CU's 160 KiB)
constexpr int RT = 4;             // row tiles per wave
constexpr int NLAYERS = 8;        // final_layer, conv 0..5, final_conv
}  // namespace tail

struct TailP {
  const float* x;        // residual stream rows [nseq * T][d] fp32, row-major
  int T, d, C, nblk;     // frames per sequence, 256, 104, frame blocks per sequence
  const h16_t* w;        // packed weight operands (tail_pack_kernel)
  const float* bias;     // [NLAYERS][256] fp32, zero beyond each layer's real output count
  float* out;            // [nseq][T][C] fp32
  int64_t woff[tail::NLAYERS];
};

// Pack one layer: dst[(kc * nt + t) * 2 + {hi, lo}][lane][8] with lane (i = lane & 15, g = lane >> 4) holding
// W[co = t*16 + i][k = kc*32 + g*8 .. +8], k = tap * cpin + ci -> src[co][ci][tap] (Conv1d layout [Co][Ci][taps]; a Linear is taps = 1),
// zero beyond the real Co / Ci.
__global__ __launch_bounds__(256) void tail_pack_kernel(const float* __restrict__ src, int Co, int Ci, int taps, int cpin, int nt, int kcs,
                                                        h16_t* __restrict__ dst) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (kc, t, lane)
  if (q >= (int64_t)kcs * nt * 64) return;
  const int lane = (int)(q & 63), t = (int)((q >> 6) % nt), kc = (int)((q >> 6) / nt);
  const int co = t * 16 + (lane & 15), k0 = kc * 32 + (lane >> 4) * 8;
  h16_t hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e, tap = k / cpin, ci = k - tap * cpin;
    const float v = (co < Co && ci < Ci && tap < taps) ? src[((int64_t)co * Ci + ci) * taps + tap] : 0.f;
    hi[e] = (h16_t)v;
    lo[e] = (h16_t)(v - (float)hi[e]);
  }
  h16_t* o = dst + ((int64_t)(kc * nt + t) * 2) * 512 + lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = hi[e]; o[512 + e] = lo[e]; }
}

// One wave = rows [half*64, +64) (4 row tiles) x TPG 16-column tiles of its column group.  A layer is a GEMM over K = taps x input
// channels, taken in slices of the 128 channels an LDS row holds (tail_mma), and an epilogue (tail_epi); 256-channel operands --
// the residual rows in front of final_layer, conv 0's output in front of conv 1 -- pass through the LDS in two halves while the
// consumer's accumulators stay in registers.
//   tail_mma: acc += in[rows - shift(tap)][coff_l .. +128) . W[cols][tap * CPW + coff_w + ..]
//     CPW: input channels per tap in the packed weights' k index; NTW: 16-column tiles per k-chunk in the packed weights;
//     t0: first tile of this wave; kcoff: k-chunk offset (32 channels each) of the slice inside a tap
template <int TAPS, int DIL, int CPW, int NTW, int TPG>
__device__ __forceinline__ void tail_mma(f32x4 (&acc)[TPG][tail::RT], const char* in, const h16_t* __restrict__ wl, int t0, int kcoff,
                                         int half, int lane) {
  constexpr int CL = tail::CL, RT = tail::RT;
  const int l15 = lane & 15, g = lane >> 4;
  const h16_t* wq = wl + ((int64_t)t0 * 2) * 512 + lane * 8;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int shift = (TAPS - 1 - tap) * DIL;
#pragma unroll 2
    for (int i = 0; i < CL / 32; ++i) {
      const int kc = tap * (CPW / 32) + kcoff + i;   // k-chunk of the packed weights
      h16x8 wh[TPG], wlo[TPG];
#pragma unroll
      for (int t = 0; t < TPG; ++t) {
        wh[t] = *reinterpret_cast<const h16x8*>(wq + ((int64_t)kc * NTW * 2 + t * 2) * 512);
        wlo[t] = *reinterpret_cast<const h16x8*>(wq + ((int64_t)kc * NTW * 2 + t * 2 + 1) * 512);
      }
      const int j = i * 4 + g;                       // 16-byte chunk of the LDS row: channels i*32 + g*8 .. +8 of the slice
      h16x8 ah[RT], al[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = tail::GUARD + half * (16 * RT) + rt * 16 + l15 - shift;
        const char* rp = in + row * (4 * CL) + ((j ^ (row & 15)) << 4);
        ah[rt] = *reinterpret_cast<const h16x8*>(rp);
        al[rt] = *reinterpret_cast<const h16x8*>(rp + 2 * CL);
      }
#pragma unroll
      for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[t][rt] = A2P_MFMA16(wh[t], ah[rt], acc[t][rt]);
          acc[t][rt] = A2P_MFMA16(wh[t], al[rt], acc[t][rt]);
          acc[t][rt] = A2P_MFMA16(wlo[t], ah[rt], acc[t][rt]);
        }
    }
  }
}

//   tail_epi: bias, leaky ReLU, skip average with the layer's input, then (hi, lo) rows of `outb` at column c0l + ..., or HBM.
//     c0: first output column of this wave in the layer's numbering (bias, skip input, HBM); c0l: the same inside the LDS row written
template <int TPG, bool LRELU, bool SKIP, bool FIRST, bool TOGLOBAL>
__device__ __forceinline__ void tail_epi(const TailP& p, f32x4 (&acc)[TPG][tail::RT], const char* in, char* outb, const float* __restrict__ bias,
                                         int c0, int c0l, int half, int lane, int f0, int seq) {
  constexpr int CL = tail::CL, RT = tail::RT;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < TPG; ++t) {
    const int co = c0 + t * 16 + 4 * g, col = c0l + t * 16 + 4 * g;
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int rl = half * (16 * RT) + rt * 16 + l15, row = tail::GUARD + rl, f = f0 + rl;   // f: frame of the sequence (may be < 0 or >= T)
      f32x4 y = acc[t][rt] + b;
      if constexpr (LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = y[e] > 0.f ? y[e] : 0.2f * y[e];
      }
      if constexpr (SKIP) {   // (in[f] + y) / 2: the input row of the same frame, hi + lo
        const char* rp = in + row * (4 * CL) + (((co >> 3) ^ (row & 15)) << 4) + (co & 7) * 2;
        const h16x4 ih = *reinterpret_cast<const h16x4*>(rp), il = *reinterpret_cast<const h16x4*>(rp + 2 * CL);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (((float)ih[e] + (float)il[e]) + y[e]) * 0.5f;
      }
      if constexpr (FIRST) {
        if (f < 0) y = f32x4{0.f, 0.f, 0.f, 0.f};   // F.pad(output, [receptive_field - 1, 0]) (model/diffusion.py:215)
      }
      if constexpr (TOGLOBAL) {
        if (rl >= tail::HALO && f < p.T && co < p.C) {   // C = 104 is a multiple of 4: a lane's four columns are in or out together
          *reinterpret_cast<f32x4*>(p.out + ((int64_t)seq * p.T + f) * p.C + co) = y;
        }
      } else {
        h16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = (h16_t)y[e];
          lo[e] = (h16_t)(y[e] - (float)hi[e]);
        }
        char* wp = outb + row * (4 * CL) + (((col >> 3) ^ (row & 15)) << 4) + (col & 7) * 2;
        *reinterpret_cast<h16x4*>(wp) = hi;
        *reinterpret_cast<h16x4*>(wp + 2 * CL) = lo;
      }
    }
  }
}

template <int TPG>
__device__ __forceinline__ void tail_zero(f32x4 (&acc)[TPG][tail::RT]) {
#pragma unroll
  for (int t = 0; t < TPG; ++t)
#pragma unroll
    for (int rt = 0; rt < tail::RT; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// a 104 -> 104 layer (conv 2..5: three taps, skip average; final_conv: one tap, to HBM) from one buffer into the other
template <int TAPS, int DIL, bool LRELU, bool SKIP, bool TOGLOBAL>
__device__ __forceinline__ void tail_layer104(const TailP& p, const char* in, char* outb, const h16_t* __restrict__ wl, const float* __restrict__ bias,
                                              int half, int ng, int lane, int f0, int seq) {
  f32x4 acc[2][tail::RT];
  tail_zero(acc);
  tail_mma<TAPS, DIL, tail::CL, 8, 2>(acc, in, wl, ng * 2, 0, half, lane);
  tail_epi<2, LRELU, SKIP, false, TOGLOBAL>(p, acc, in, outb, bias, ng * 32, ng * 32, half, lane, f0, seq);
}

__global__ __launch_bounds__(512, 1) void pose_tail_kernel(const TailP p) {
  using namespace tail;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  char* const bufA = smem;
  char* const bufB = smem + BUF;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = wid >> 2, ng = wid & 3;
  const int seq = blockIdx.x / p.nblk, blk = blockIdx.x - seq * p.nblk;
  const int f0 = blk * TB - HALO;   // frame of computed row 0
  const h16_t* w = p.w;
  const float* bs = p.bias;

  // guard rows of both buffers: zeros (they feed only rows no output depends on; zeros keep those rows finite)
  for (int i = tid * 16; i < GUARD * 4 * CL; i += 512 * 16) {
    *reinterpret_cast<uint4*>(bufA + i) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(bufB + i) = make_uint4(0, 0, 0, 0);
  }
  // ---- final_layer (model/diffusion.py:397): K = 256 residual-stream channels, through bufB in two halves of 128 ----
  f32x4 acc[2][RT];
  tail_zero(acc);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();   // every wave is done reading the first half
    // residual rows of the 128 frames, channels [128h, +128) -> (hi, lo) rows of bufB: 16 threads per row, 8 channels each
    for (int r0 = 0; r0 < ROWS; r0 += 32) {
      const int rl = r0 + (tid >> 4), c = (tid & 15) * 8, f = f0 + rl, row = GUARD + rl;
      f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (f >= 0 && f < p.T) {
        const float* xp = p.x + ((int64_t)seq * p.T + f) * p.d + h * CL + c;
        v0 = *reinterpret_cast<const f32x4*>(xp);
        v1 = *reinterpret_cast<const f32x4*>(xp + 4);
      }
      h16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (h16_t)v0[e]; lo[e] = (h16_t)(v0[e] - (float)hi[e]);
        hi[4 + e] = (h16_t)v1[e]; lo[4 + e] = (h16_t)(v1[e] - (float)hi[4 + e]);
      }
      char* wp = bufB + row * (4 * CL) + (((c >> 3) ^ (row & 15)) << 4);
      *reinterpret_cast<h16x8*>(wp) = hi;
      *reinterpret_cast<h16x8*>(wp + 2 * CL) = lo;
    }
    __syncthreads();
    tail_mma<1, 1, 256, 8, 2>(acc, bufB, w + p.woff[0], ng * 2, h * 4, half, lane);
  }
  tail_epi<2, false, false, true, false>(p, acc, nullptr, bufA, bs + 0 * 256, ng * 32, ng * 32, half, lane, f0, seq);
  __syncthreads();
  // ---- conv 0 (104 -> 256, dilation 1) and conv 1 (256 -> 104, dilation 2): conv 0's 256 output channels pass through bufB in two
  // halves, conv 1 accumulates over each half while it is there ----
  tail_zero(acc);
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    f32x4 a0[2][RT];
    tail_zero(a0);
    tail_mma<3, 1, CL, 16, 2>(a0, bufA, w + p.woff[1], h * 8 + ng * 2, 0, half, lane);
    if (h) __syncthreads();   // conv 1 is done reading the first half of conv 0's output
    tail_epi<2, true, false, false, false>(p, a0, nullptr, bufB, bs + 1 * 256, h * 128 + ng * 32, ng * 32, half, lane, f0, seq);
    __syncthreads();
    tail_mma<3, 2, 256, 8, 2>(acc, bufB, w + p.woff[2], ng * 2, h * 4, half, lane);
  }
  __syncthreads();            // conv 0 is done reading bufA (its input) in every wave
  tail_epi<2, true, false, false, false>(p, acc, nullptr, bufA, bs + 2 * 256, ng * 32, ng * 32, half, lane, f0, seq);
  __syncthreads();
  // ---- conv 2..5 (104 -> 104, dilations 3 1 2 3, skip average), final_conv ----
  tail_layer104<3, 3, true, true, false>(p, bufA, bufB, w + p.woff[3], bs + 3 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer104<3, 1, true, true, false>(p, bufB, bufA, w + p.woff[4], bs + 4 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer104<3, 2, true, true, false>(p, bufA, bufB, w + p.woff[5], bs + 5 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer104<3, 3, true, true, false>(p, bufB, bufA, w + p.woff[6], bs + 6 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer104<1, 1, false, false, true>(p, bufA, nullptr, w + p.woff[7], bs + 7 * 256, half, ng, lane, f0, seq);   // final_conv -> HBM
}
#pragma clang fp contract(fast)
