// Fused output tail of the body model (round 4): final_layer (model/diffusion.py:397) + the six dilated convolutions of
// _run_single_pose_conv (:214-224: k = 3, dilations 1 2 3 1 2 3, 104 -> 256 -> 104 -> 104 ..., leaky ReLU 0.2, the
// (output[..., -len(y):] + y) / 2 skip connections, left padding of receptive_field - 1 = 24 zero frames) + final_conv (:401), as
// ONE kernel whose activations never leave the LDS.  Rounds 2-3 ran this as 9 split-operand GEMM launches of 23-26 us each
// (12 % of the body step for 0.6 % of its FLOPs: every launch is one HBM round trip of a [19200 x 128|256] activation).
//
// A workgroup (8 waves: 2 row halves x 4 column groups) owns 64 output frames of one sequence.  The convolutions are causal in
// the reference's right-aligned indexing -- y[f] = sum_k W[:, :, k] . in[f - (2 - k) * dilation] -- so the block recomputes a halo
// of 24 frames on its left (+ 8 slack rows: 96 rows = six 16-row MFMA tiles) from the residual stream and no workgroup ever
// waits for another.  Frames before the start of the sequence enter as zeros behind final_layer (F.pad), exactly like the
// reference; deeper layers are computed from them, not zeroed.
//
// Arithmetic: the exact island of the 16-bit modes -- every operand is a (hi, lo) pair of 16-bit values
// and a . w = a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32 accumulators (the dropped a_lo w_lo is 2^-22 relative with IEEE half).
// Activations live in LDS as [row][hi: CP | lo: CP] 16-bit rows, 16-byte chunks XOR-swizzled by (row & 15) (conflict-free
// ds_read_b128 of 16 consecutive rows, also when a tap shifts the rows); weights stream from L2 as ready-made MFMA operands
// ([layer][k-chunk][16-column tile][hi | lo][64 lanes][8]: one global_load_dwordx4 of a wave = one operand), packed once at
// a2p_finalize_weights (tail_pack_kernel).
//
// Work per workgroup: 15.6 k MFMA 16x16x32 (three per product) + 3.4 MB of weight operands through L2; 300 workgroups at the
// bench shape (32 sequences x 600 frames) = two rounds of the 256 CUs.
#pragma once
#include "a2p_common.h"

#pragma clang fp contract(off)

namespace tail {
constexpr int TB = 64;            // output frames per workgroup
constexpr int HALO = 32;          // rows computed in front of them (24 needed + 8 slack)
constexpr int ROWS = TB + HALO;   // 96 = 2 halves x 3 row tiles of 16
constexpr int GUARD = 8;          // zero rows in front of the computed ones: taps of the first (never used) rows stay inside the buffer
constexpr int LROWS = ROWS + GUARD;
constexpr int CA = 128, CB = 256; // channels per row of the two LDS buffers (104 padded to 128; 256)
constexpr int BUFA = LROWS * 4 * CA, BUFB = LROWS * 4 * CB;   // bytes: [row][hi CP | lo CP] 16-bit
constexpr int NLAYERS = 8;        // final_layer, conv 0..5, final_conv
}  // namespace tail

struct TailLayer {   // one layer of the fused tail as a GEMM over [rows] x [K = taps * CPIN] x [16 * NT columns]
  int cpin, taps, dil, nt, kc;    // input channels per row (padded), taps, dilation, 16-column output tiles (padded), k-chunks of 32
  int64_t woff;                   // element offset of the layer's operands in the packed stream
};

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

// One layer for one wave: rows [half*48, +48) x the TPG tiles of its column group.
//   in : LDS buffer with CPIN channels per row (taps read rows shifted by (2 - tap) * DIL); out: LDS buffer with CPOUT channels, or
//   global memory (TOGLOBAL: final_conv).  FIRST: final_layer -- no activation, frames before the sequence start become zeros.
template <int CPIN, int TAPS, int DIL, int NT, int CPOUT, bool LRELU, bool SKIP, bool FIRST, bool TOGLOBAL>
__device__ __forceinline__ void tail_layer(const TailP& p, const char* in, char* outb, const h16_t* __restrict__ wl, const float* __restrict__ bias,
                                           int half, int ng, int lane, int f0, int seq) {
  constexpr int TPG = NT / 4;               // 16-column tiles per column group
  constexpr int KC = TAPS * CPIN / 32;      // k-chunks
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 acc[TPG][3];
#pragma unroll
  for (int t = 0; t < TPG; ++t)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const h16_t* wq = wl + ((int64_t)ng * TPG * 2) * 512 + lane * 8;   // this wave's first operand of k-chunk 0
#pragma unroll 2
  for (int kc = 0; kc < KC; ++kc) {
    // weights: TPG tiles x (hi, lo), contiguous in the stream
    h16x8 wh[TPG], wlo[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      wh[t] = *reinterpret_cast<const h16x8*>(wq + ((int64_t)kc * NT * 2 + t * 2) * 512);
      wlo[t] = *reinterpret_cast<const h16x8*>(wq + ((int64_t)kc * NT * 2 + t * 2 + 1) * 512);
    }
    // activations: k = kc*32 + g*8 -> (tap, channel chunk); the lane's row of row tile rt, shifted by the tap
    const int k = kc * 32 + g * 8, tap = k / CPIN, j = (k - tap * CPIN) >> 3;
    const int shift = (TAPS - 1 - tap) * DIL;
    h16x8 ah[3], al[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      const int row = tail::GUARD + half * 48 + rt * 16 + l15 - shift;
      const char* rp = in + row * (4 * CPIN) + ((j ^ (row & 15)) << 4);
      ah[rt] = *reinterpret_cast<const h16x8*>(rp);
      al[rt] = *reinterpret_cast<const h16x8*>(rp + 2 * CPIN);
    }
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        acc[t][rt] = A2P_MFMA16(wh[t], ah[rt], acc[t][rt]);
        acc[t][rt] = A2P_MFMA16(wh[t], al[rt], acc[t][rt]);
        acc[t][rt] = A2P_MFMA16(wlo[t], ah[rt], acc[t][rt]);
      }
  }
  // epilogue: lane holds columns co = tile*16 + 4g + {0..3} of frame row l15 of every row tile
#pragma unroll
  for (int t = 0; t < TPG; ++t) {
    const int co = (ng * TPG + t) * 16 + 4 * g;
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      const int rl = half * 48 + rt * 16 + l15, row = tail::GUARD + rl, f = f0 + rl;   // f: frame of the sequence (may be < 0 or >= T)
      f32x4 y = acc[t][rt] + b;
      if constexpr (LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = y[e] > 0.f ? y[e] : 0.2f * y[e];
      }
      if constexpr (SKIP) {   // (in[f] + y) / 2: the input row of the same frame, hi + lo
        const char* rp = in + row * (4 * CPIN) + (((co >> 3) ^ (row & 15)) << 4) + (co & 7) * 2;
        const h16x4 ih = *reinterpret_cast<const h16x4*>(rp), il = *reinterpret_cast<const h16x4*>(rp + 2 * CPIN);
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
        char* wp = outb + row * (4 * CPOUT) + (((co >> 3) ^ (row & 15)) << 4) + (co & 7) * 2;
        *reinterpret_cast<h16x4*>(wp) = hi;
        *reinterpret_cast<h16x4*>(wp + 2 * CPOUT) = lo;
      }
    }
  }
}

__global__ __launch_bounds__(512, 1) void pose_tail_kernel(const TailP p) {
  using namespace tail;
  __shared__ __attribute__((aligned(16))) char smem[BUFA + BUFB];
  char* const bufA = smem;          // 128 channels per row
  char* const bufB = smem + BUFA;   // 256 channels per row (also used with 128-channel rows)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = wid >> 2, ng = wid & 3;
  const int seq = blockIdx.x / p.nblk, blk = blockIdx.x - seq * p.nblk;
  const int f0 = blk * TB - HALO;   // frame of computed row 0

  // guard rows of both buffers: zeros (they feed only rows no output depends on; zeros keep those rows finite)
  for (int i = tid * 16; i < GUARD * 4 * CA; i += 512 * 16) *reinterpret_cast<uint4*>(bufA + i) = make_uint4(0, 0, 0, 0);
  for (int i = tid * 16; i < GUARD * 4 * CB; i += 512 * 16) *reinterpret_cast<uint4*>(bufB + i) = make_uint4(0, 0, 0, 0);
  // residual-stream rows of the 96 frames -> (hi, lo) rows of bufB (256 channels): 32 threads per row, 8 channels each
  for (int r0 = 0; r0 < ROWS; r0 += 16) {
    const int rl = r0 + (tid >> 5), c = (tid & 31) * 8, f = f0 + rl, row = GUARD + rl;
    f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (f >= 0 && f < p.T) {
      const float* xp = p.x + ((int64_t)seq * p.T + f) * p.d + c;
      v0 = *reinterpret_cast<const f32x4*>(xp);
      v1 = *reinterpret_cast<const f32x4*>(xp + 4);
    }
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (h16_t)v0[e]; lo[e] = (h16_t)(v0[e] - (float)hi[e]);
      hi[4 + e] = (h16_t)v1[e]; lo[4 + e] = (h16_t)(v1[e] - (float)hi[4 + e]);
    }
    char* wp = bufB + row * (4 * CB) + (((c >> 3) ^ (row & 15)) << 4);
    *reinterpret_cast<h16x8*>(wp) = hi;
    *reinterpret_cast<h16x8*>(wp + 2 * CB) = lo;
  }
  __syncthreads();
  const h16_t* w = p.w;
  const float* bs = p.bias;
  //          CPIN TAPS DIL NT CPOUT  LRELU  SKIP   FIRST  TOGLOBAL
  tail_layer<CB, 1, 1, 8, CA, false, false, true, false>(p, bufB, bufA, w + p.woff[0], bs + 0 * 256, half, ng, lane, f0, seq);    // final_layer
  __syncthreads();
  tail_layer<CA, 3, 1, 16, CB, true, false, false, false>(p, bufA, bufB, w + p.woff[1], bs + 1 * 256, half, ng, lane, f0, seq);   // 104 -> 256, dilation 1
  __syncthreads();
  tail_layer<CB, 3, 2, 8, CA, true, false, false, false>(p, bufB, bufA, w + p.woff[2], bs + 2 * 256, half, ng, lane, f0, seq);    // 256 -> 104, dilation 2
  __syncthreads();
  tail_layer<CA, 3, 3, 8, CA, true, true, false, false>(p, bufA, bufB, w + p.woff[3], bs + 3 * 256, half, ng, lane, f0, seq);     // dilation 3, skip
  __syncthreads();
  tail_layer<CA, 3, 1, 8, CA, true, true, false, false>(p, bufB, bufA, w + p.woff[4], bs + 4 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer<CA, 3, 2, 8, CA, true, true, false, false>(p, bufA, bufB, w + p.woff[5], bs + 5 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer<CA, 3, 3, 8, CA, true, true, false, false>(p, bufB, bufA, w + p.woff[6], bs + 6 * 256, half, ng, lane, f0, seq);
  __syncthreads();
  tail_layer<CA, 1, 1, 8, CA, false, false, false, true>(p, bufA, nullptr, w + p.woff[7], bs + 7 * 256, half, ng, lane, f0, seq);  // final_conv -> HBM
}
#pragma clang fp contract(fast)
