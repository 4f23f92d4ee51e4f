// liba2p_hip.so -- host orchestration + C ABI (include/a2p_hip.h) of the MI355X-native
// audio2photoreal sampling hot path.  One translation unit: device kernels live in the headers.
//
// Data layout in HBM (all owned by the context, sized once for max_batch):
//   residual stream x        fp32 [N*T, d]           N = sequences (2B under classifier-free guidance)
//   activations xn/xr/q/k/o  T    [N*T, d|2d]        T = fp32 (parity mode) or bf16 (throughput mode)
//   V^T (self attention)     T    [N][d][T_ld]       transposed by the producing GEMM epilogue
//   audio K cache            T    [B+1][S_ld][L*d]   slot 0 = batch-invariant unconditional branch
//   audio V^T cache          T    [B+1][L*d][S_ld]
//   keyframe K / V^T cache   T    [B+1][64][L*d] / [B+1][L*d][64]     (pose)
//   FiLM scale|shift         fp32 [N][L][F][2d]      regenerated every step from the time path
//   time-token K / V tail    fp32 [B*2][L*d]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <tuple>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/a2p_hip.h"
#include "kernels_attn.h"
#include "kernels_attn2.h"
#include "kernels_attn3.h"
#include "kernels_chain.h"
#include "kernels_chain4.h"
#include "kernels_gemm.h"
#include "kernels_misc.h"
#include "kernels_small.h"
#include "kernels_tail.h"

static thread_local char g_err[1024] = "";
static void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
#define HIPCHK(x)                                                                 \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      set_err("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      return A2P_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)
#define CHK(x)            \
  do {                    \
    int r_ = (x);         \
    if (r_ < 0) return r_; \
  } while (0)
#define ARG(cond, ...)       \
  do {                       \
    if (!(cond)) {           \
      set_err(__VA_ARGS__);  \
      return A2P_ERR_ARG;    \
    }                        \
  } while (0)

struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
  bool pooled = false;  // carved from the context's arena: released with it, not individually
  float* f() const { return reinterpret_cast<float*>(p); }
};

// Device memory of a context comes from a few 512 MiB slabs instead of ~700 individual hipMallocs (one per parameter,
// per packed copy, per workspace): one contiguous range per context, context creation ~10x fewer driver calls.
// (Measured: no effect on kernel time -- A2P_NO_ARENA=1 restores per-buffer hipMalloc for A/B runs.)
struct Arena {
  std::vector<void*> slabs;
  char* cur = nullptr;
  size_t left = 0;
  static constexpr size_t kSlab = (size_t)512 << 20;
};
static thread_local Arena* g_arena = nullptr;  // set while a context allocates its long-lived buffers

// Zero-fill of a fresh allocation, COMPLETE before the call returns.  (Rounds 1-3 used hipMemset: it is enqueued on the null stream
// and may return before it has run, and the null stream is not ordered against the NON-BLOCKING streams a caller may hand to the
// entry points (torch.cuda.Stream): the first kernels that wrote such a buffer -- weight repacks at finalize time, the guide's
// conv stack -- raced with the zero-fill and could find part of their output zeroed afterwards.  Found in round 4 as "the first
// process on a fresh box samples different keyframes than every later one" under bench.py --pipeline's two-stream schedule.)
static int zero_fill_now(void* p, size_t bytes) {
  HIPCHK(hipMemsetAsync(p, 0, bytes, nullptr));
  HIPCHK(hipStreamSynchronize(nullptr));
  return 0;
}
static int buf_alloc_tmp(Buf& b, size_t bytes) {  // short-lived scratch of the unit / setup entry points
  if (b.p && !b.pooled) (void)hipFree(b.p);
  b.p = nullptr;
  if (bytes == 0) bytes = 16;
  HIPCHK(hipMalloc(&b.p, bytes));
  CHK(zero_fill_now(b.p, bytes));
  b.bytes = bytes;
  b.pooled = false;
  return 0;
}
static int buf_alloc(Buf& b, size_t bytes) {
  Arena* a = g_arena;
  if (!a) return buf_alloc_tmp(b, bytes);
  if (bytes == 0) bytes = 16;
  if (b.p && b.pooled && b.bytes >= bytes) {  // re-finalisation after a weight update: same slot
    // (the slot may still be read by work the caller enqueued earlier on ITS stream: let the device drain first)
    HIPCHK(hipDeviceSynchronize());
    CHK(zero_fill_now(b.p, bytes));
    return 0;
  }
  if (b.p && !b.pooled) (void)hipFree(b.p);
  const size_t align = bytes >= ((size_t)2 << 20) ? ((size_t)2 << 20) : 256;
  size_t pad = (align - (reinterpret_cast<uintptr_t>(a->cur) & (align - 1))) & (align - 1);
  if (!a->cur || pad + bytes > a->left) {
    const size_t slab = bytes + align > Arena::kSlab ? bytes + align : Arena::kSlab;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, slab));
    a->slabs.push_back(p);
    a->cur = reinterpret_cast<char*>(p);
    a->left = slab;
    pad = (align - (reinterpret_cast<uintptr_t>(a->cur) & (align - 1))) & (align - 1);
  }
  b.p = a->cur + pad;
  a->cur += pad + bytes;
  a->left -= pad + bytes;
  CHK(zero_fill_now(b.p, bytes));
  b.bytes = bytes;
  b.pooled = true;
  return 0;
}
static void buf_free(Buf& b) {
  if (b.p && !b.pooled) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  b.pooled = false;
}
struct ArenaScope {  // RAII: route buf_alloc to a context's arena for the duration of an entry point
  Arena* prev;
  explicit ArenaScope(Arena* a) : prev(g_arena) { g_arena = a; }
  ~ArenaScope() { g_arena = prev; }
};
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// Run-time switches of the forward (A/B runs, tests, diagnostics), read from the environment ONCE per context -- and again when
// the host asks (a2p_reload_env; the Python mirror calls it when an A2P_* variable changed) -- instead of ~10 getenv calls per forward.
struct A2POpts {
  int kv_cached = 0;        // A2P_KV_CACHED=1: default cache policy for the cached audio K/V (A/B)
  int no_chain = 0;         // A2P_NO_CHAIN=1: per-op kernels instead of the chain kernels
  int chain_nw = 0;         // A2P_CHAIN_NW=4|8: force the generation-1 workgroup shape (0: measured per box)
  int chain_tune = 0;       // A2P_CHAIN_TUNE=1: measure the 4- vs 8-wave chain workgroup shape in situ (round 3's default); 0: 8 waves
  int chain_mt = 0;         // A2P_CHAIN_MT=n: force the panel height 16*n (also takes the chain path below 960 rows)
  int chain_no_mix = 0;     // A2P_CHAIN_NO_MIX=1: uniform panel heights
  int tune_verbose = 0;     // A2P_TUNE_VERBOSE=1: print the per-box shape decision
  int side_join = 3;        // A2P_SIDE_JOIN: where the main stream joins the side stream (1 PRE, 2 self attention, 3 MID)
  int x_rowmajor = 0;       // A2P_CHAIN_X_ROWMAJOR=1: residual stream row-major between chain kernels (A/B)
  int no_side_stream = 0;   // A2P_NO_SIDE_STREAM=1: time path on the main stream
  int side_early_join = 0;  // A2P_SIDE_EARLY_JOIN=1: side stream without overlap (diagnostic)
  int no_shared_half = 0;   // A2P_NO_SHARED_HALF=1: layer 0 computed for both guidance halves
  int no_ksplit = 0;        // A2P_NO_KSPLIT=1: small forwards use attn_kernel instead of the key-split attention (A/B)
  int ksplit_nw = 0;        // A2P_KSPLIT_NW=4|8: waves of the key-split attention
  int ksplit_qt = 0;        // A2P_KSPLIT_QT=1|2: 16-query tiles per wave of the key-split attention
  int force_ksplit = 0;     // A2P_ATTN_KSPLIT=1: every 16-bit attention launch takes the key-split kernel (tests)
  int chain_v = 0;          // A2P_CHAIN_V=4: tall chain kernels (kernels_chain4.h) wherever their contract holds; A2P_CHAIN_V=1: never; 0: the
                            // faster of the two families ON THIS BOX, measured during the first forwards of a size (chain_pick_family)
  int attn3 = 1;            // A2P_ATTN3: 1 (default) = the one-wave-per-SIMD kernel with the ISA-level tile body (kernels_attn3.h) where it wins
                            // (head_dim 64, >= 160 queries, and >= 1024 keys or at most one workgroup per CU: launch_attn); 2 = wherever it is legal
                            // (tests); 0 = never (A/B)
  int attn2 = 0;            // A2P_ATTN2=1: the query-split 16-bit attention launches take attn2_kernel (kernels_attn2.h: one 8-wave workgroup per CU,
                            // 48 + 32 queries per SIMD, unit-level software pipeline); 0: attn_kernel
  int no_fused_kf = 0;      // A2P_NO_FUSED_KF=1: body model: MID2 | keyframe attention | POST as three launches instead of one (A/B, tests)
  int time_table = 1000;    // A2P_TIME_TABLE: rows of the time-MLP table (a2p_ctx::tct_table) built at a2p_finalize_weights; 0 = compute the time MLP in every forward
  int no_fused_in = 0;      // A2P_NO_FUSED_IN=1: face model: pack_input_split3 + gemm_kernel + the gen-1 PRE kernel instead of chain4_kernel<MT, CHAIN_IN> (A/B, tests)
  int no_fused_final = 0;   // A2P_NO_FUSED_FINAL=1: face model: final_layer behind the last tall POST kernel as split3_kernel + gemm_kernel instead of inside it (A/B, tests)
  int graph = 0;            // A2P_GRAPH=1: non-chain forwards replay a captured graph instead of stream launches (measured: same GPU
                            // time per step -- the launches are not host-bound -- at a tenth of the host time; off by default)
  int no_small = 0;         // A2P_NO_SMALL=1: per-op kernels for forwards below 960 rows instead of kernels_small.h
  int chain_rows = 1100;    // A2P_CHAIN_ROWS=n: forwards of at least n rows take the chain kernels.  Measured against the small-forward kernels
                            // (profiles/r03_ksplit_ab.txt): 960 rows (2 x 2 x 240) small 1192 vs chain 1002 steps/s; 1200 rows (2 x 600 frames, 2000
                            // keys) 761 vs 878; 1440 (2 x 3 x 240) 939 vs 1022; 1920: 798 vs 1020; 2400: 488 vs 870
};
static void load_opts(A2POpts& o) {
  auto flag = [](const char* n) { return getenv(n) != nullptr ? 1 : 0; };
  auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
  o.kv_cached = flag("A2P_KV_CACHED"); o.no_chain = flag("A2P_NO_CHAIN");
  o.chain_nw = num("A2P_CHAIN_NW", 0); o.chain_mt = num("A2P_CHAIN_MT", 0);
  o.chain_no_mix = flag("A2P_CHAIN_NO_MIX"); o.chain_tune = flag("A2P_CHAIN_TUNE"); o.tune_verbose = flag("A2P_TUNE_VERBOSE"); o.side_join = num("A2P_SIDE_JOIN", 3);
  o.x_rowmajor = flag("A2P_CHAIN_X_ROWMAJOR"); o.no_side_stream = flag("A2P_NO_SIDE_STREAM");
  o.side_early_join = flag("A2P_SIDE_EARLY_JOIN"); o.no_shared_half = flag("A2P_NO_SHARED_HALF");
  o.graph = flag("A2P_GRAPH");
  o.no_ksplit = flag("A2P_NO_KSPLIT"); o.force_ksplit = flag("A2P_ATTN_KSPLIT"); o.ksplit_nw = num("A2P_KSPLIT_NW", 0); o.ksplit_qt = num("A2P_KSPLIT_QT", 0);
  o.no_fused_kf = flag("A2P_NO_FUSED_KF"); o.no_fused_final = flag("A2P_NO_FUSED_FINAL"); o.no_fused_in = flag("A2P_NO_FUSED_IN"); o.time_table = num("A2P_TIME_TABLE", 1000); o.attn2 = num("A2P_ATTN2", 0); o.attn3 = num("A2P_ATTN3", 1); o.chain_v = num("A2P_CHAIN_V", 0);
  o.no_small = flag("A2P_NO_SMALL"); o.chain_rows = num("A2P_CHAIN_ROWS", 1100);
}

struct a2p_ctx {
  a2p_config cfg;
  A2POpts opt;
  Arena arena;
  bool use_arena = true;
  int d, H, DH, L, C, Cpad, ff, F, Fc, FcPad, Kd, KdPad, Tmax, Tld, S0max, Sld, Bmax, Nmax, KFmax;
  bool bf16, pose;
  // 16-bit modes: the operand classes that carry the error of the sampling LOOP's return value stay exact fp32 (error budget in
  // profiles/r03_error_budget*.json: final_layer's operand rows alone are 3.05e-3 of the 3.11e-3 loop error of the face model,
  // the dilated conv tail 7e-4 of the body model's 8.2e-4): input_projection, final_layer and the pose conv tail run as
  // exact-fp32 MFMA GEMMs on fp32 buffers (< 3 % of a step's FLOPs).  A2P_TAIL16=1 restores the all-16-bit path for A/B runs.
  bool tail32 = false;
  // ... and by default not as fp32 MFMA (1/16 of the 16-bit rate: -14 % steps/s for the body model) but as SPLIT-OPERAND 16-bit
  // GEMMs (kernels_misc.h split3_kernel: hi/lo pairs, three products per element over a 3x longer contraction; the dropped lo*lo
  // term is 2^-22 relative).  A2P_TAIL_F32=1 keeps the fp32 MFMA islands for A/B runs.
  bool tail_x3 = false;
  size_t esz;
  int64_t rows_cap, conv_rows;
  std::map<std::string, Buf> w;   // fp32 parameters by reference state_dict key
  std::map<std::string, Buf> wt;  // compute-dtype copies [N, Kpad]
  bool finalized = false, prepared = false;
  int rope_npos = 0;
  Buf rope_cs, rope_cst, time_freq, film_w, film_b, tct_w, tct_b;
  Buf tct_table;         // [tct_rows][3d]: the time MLP's outputs (time cond | token 0 | token 1) for every timestep value 0 .. tct_rows-1, built at a2p_finalize_weights by the
  int tct_rows = 0;      // same kernels that compute them per forward otherwise (same bits); A2P_TIME_TABLE=rows (default 1000 = the reference's diffusion_steps; 0: off)
  Buf cak_w32, cak_b, cav_w32, cav_b, cak_wt, cav_wt;
  Buf ca2k_wt, ca2v_wt, ca2k_b, ca2v_b;
  Buf conv_wt[7];
  Buf tail_w, tail_b;                  // fused output tail of the body model (kernels_tail.h): packed MFMA weight operands, [8][256] biases
  int64_t tail_woff[8] = {};
  bool tail_fused = false;
  int64_t attn3_launches = 0;          // launches of attn3_kernel (a2p_debug_read "attn3_launches")
  int64_t ch4_launches = 0;            // launches of the tall chain kernels (a2p_debug_read "chain4_launches")
  int64_t fin_fused_launches = 0;      // ... of those, last-layer POST kernels that computed final_layer too (a2p_debug_read "final_fused_launches")
  std::vector<Buf> ch_stream4w;        // POST streams with 256-column hidden chunks [layer]
  Buf ch_stream4_in, ch_aux_in;        // CHAIN_IN (input_projection + layer 0's PRE work in one tall kernel): stream [W_hi | W_hi | W_lo | Q|K | V], aux [bias_in 512 | 0 | bias_qkv]
  int64_t in4_launches = 0;            // launches of chain4_kernel<MT, CHAIN_IN> (a2p_debug_read "chain_in_launches")
  std::vector<Buf> ch_stream4;         // kernels_chain4.h: half-stage streams [layer*5 + kind] (CH_MID, CH_POST of the face model; empty Buf otherwise)
  std::vector<Buf> ch_stream, ch_aux;  // packed weight streams [layout * L*5 + layer*5 + kind] (layout 0: 4-wave LDS slices, 1: 8-wave; kinds: a2p_lib_run.h CH_*) / bias blocks [layer*5 + kind]
  int ch_nw = 4;                       // waves per chain workgroup of the forward being enqueued (4 or 8; chain_pick_nw)
  struct ChainTune {                   // per forward size (rows): which workgroup shape is faster ON THIS BOX, measured in situ
    int choice = 0, calls = 0;
    std::vector<std::tuple<int, hipEvent_t, hipEvent_t>> samples;
  };
  std::map<int64_t, ChainTune> ch_tune;
  std::map<int64_t, ChainTune> ch_tune4;   // per forward size: kernels_chain.h (1) or kernels_chain4.h (4), measured in situ (chain_pick_family)
  int ch_fam_mid = 4, ch_fam_post = 4;     // chain kernel families of the forward being enqueued (MID kernels, POST kernels)
  Buf hidden, kc, vtc, k2c, vt2c, slot_cond, slot_unc, slot_cfg;
  Buf nonfinite;         // device flag of a2p_check_finite (one int, OR-ed by step_tail_kernel)
  Buf clk;               // A2P_CHAIN_CLK=1: 64 chain launches x 8 blocks x {memtime, realtime} x {begin, end}
  unsigned clk_turn = 0;
  int pB = 0, pS0 = 0, pT = 0, pK = 0;
  int batch_hint = 0;    // a2p_set_batch_hint: samples of the UNSHARDED batch this context's batch is a block of (0: unknown)
  // workspaces
  Buf x, xn, xr, qk, vt, ao, hff, inpack, mo, cb[4], t3;
  Buf emb, th, tct, tvec, mt, tokn, tokr, film, ktail, vtail;
  Buf ce_pack, pooled, tmpa, tmpb, kf_pack, kf_tok;
  // timing
  int time_kind = -1;
  // captured forwards (a2p_lib_run.h run_forward)
  struct GraphKey { int pass, B, T, S0, K, small, epoch; };
  struct GraphEnt { GraphKey key; hipGraphExec_t exec; int rows; };
  std::vector<GraphEnt> graphs;
  hipStream_t gstream = nullptr;
  int graph_epoch = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  std::vector<hipEvent_t> ev_pool;
  // side stream: the per-step time path (t -> FiLM scale/shift, time-token K/V) overlaps the first projections / self attention
  hipStream_t side = nullptr;
  hipEvent_t ev_fork_pool[8] = {}, ev_join_pool[8] = {};  // rotating pairs: an event is never re-recorded while a wait on it may be pending
  unsigned ev_turn = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // the pair of the current forward

  void* offT(const Buf& b, int64_t elems) const { return reinterpret_cast<char*>(b.p) + elems * (int64_t)esz; }
  void* offT(void* p, int64_t elems) const { return reinterpret_cast<char*>(p) + elems * (int64_t)esz; }
};

// RAII: inside the scope the launch helpers (launch_gemm, launch_cast, make_wt, offT) treat the context as an fp32 one --
// the exact-fp32 islands of the 16-bit modes (a2p_ctx::tail32)
struct Fp32Scope {
  a2p_ctx* c;
  bool on, saved;
  Fp32Scope(a2p_ctx* ctx, bool enable) : c(ctx), on(enable), saved(ctx->bf16) {
    if (on) { c->bf16 = false; c->esz = 4; }
  }
  ~Fp32Scope() {
    if (on) { c->bf16 = saved; c->esz = saved ? 2 : 4; }
  }
};

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// Per-kernel timing for bench.py's roofline leg: when enabled for a kernel class, the launch goes through
// hipExtLaunchKernelGGL, which stamps the start/stop events from the dispatch packet itself (the kernel's own begin/end
// on the launch stream) -- bracketing hipEventRecord calls would add ~20-35 us of marker-packet latency per launch.
struct KernelTimer {
  a2p_ctx* c;
  bool on;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // `sub`: a finer class of the same launch (A2P_KERNEL_CHAIN_PRE.., A2P_KERNEL_POSE_TAIL): timed when either is selected
  KernelTimer(a2p_ctx* ctx, int kind, int sub = -2) : c(ctx), on(ctx && (ctx->time_kind == kind || ctx->time_kind == sub)) {
    if (on) {
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
    }
  }
  ~KernelTimer() {
    if (on) c->evs.push_back({e0, e1});
  }
};
#define A2P_LAUNCH(kt, kernel, grid, block, stream, ...)                                                        \
  do {                                                                                                          \
    if ((kt).on) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, (kt).e0, (kt).e1, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__);                           \
  } while (0)

static GemmP gemm_base(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
                       int M, int N, int K) {
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.out = out; p.ldo = ldo;
  p.M = M; p.N = N; p.K = K; p.ntaps = 1; p.epi = EPI_STORE; p.act = ACT_NONE; p.rows_per_seq = 1;
  return p;
}

template <typename T, int MT, int NB = 2>
static int gemm_dispatch(KernelTimer& kt, const GemmP& p, hipStream_t s) {
  dim3 grid((p.N + 127) / 128, (p.M + 32 * MT - 1) / (32 * MT));
#define A2P_GEMM(EPI, ACT, F32) A2P_LAUNCH(kt, (gemm_kernel<T, MT, EPI, ACT, F32, NB>), grid, 256, s, p)
  if (p.epi == EPI_FILM_RES) A2P_GEMM(EPI_FILM_RES, ACT_NONE, false);
  else if (p.epi == EPI_STORE_T && p.act == ACT_NONE) A2P_GEMM(EPI_STORE_T, ACT_NONE, false);
  else if (p.epi == EPI_CONV && p.act == ACT_LRELU && !p.out_f32) A2P_GEMM(EPI_CONV, ACT_LRELU, false);
  else if (p.epi == EPI_STORE && p.act == ACT_NONE && p.out_f32) A2P_GEMM(EPI_STORE, ACT_NONE, true);
  else if (p.epi == EPI_STORE && p.act == ACT_NONE) A2P_GEMM(EPI_STORE, ACT_NONE, false);
  else if (p.epi == EPI_STORE && p.act == ACT_GELU && !p.out_f32) A2P_GEMM(EPI_STORE, ACT_GELU, false);
  else if (p.epi == EPI_STORE && p.act == ACT_RELU && !p.out_f32) A2P_GEMM(EPI_STORE, ACT_RELU, false);   // audio front end
  else if (p.epi == EPI_STORE && p.act == ACT_RELU) A2P_GEMM(EPI_STORE, ACT_RELU, true);                   // its last conv layer in 16-bit mode
  else {
    set_err("gemm: no kernel instance for epi=%d act=%d out_f32=%d", p.epi, p.act, p.out_f32);
    return A2P_ERR_ARG;
  }
#undef A2P_GEMM
  return 0;
}

static int launch_gemm(a2p_ctx* c, const GemmP& p, hipStream_t s) {
  const int bk = c->bf16 ? 64 : 32;
  ARG(p.K % bk == 0 && p.N % 4 == 0 && p.M > 0, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  // bf16: 64x128 tiles at every shape of this path (measured on MI355X, profiles/r01_gemm_ablation.txt: 9600..38400 rows x
  // 512|1024 cols, 64x128 is 0-25% faster than 128x128 -- three co-resident blocks per CU overlap each other's
  // load / MFMA / epilogue phases).  fp32: 128x128 unless that leaves the 256 CUs under two blocks each.
  const int64_t blocks128 = (int64_t)((p.N + 127) / 128) * ((p.M + 127) / 128);
  const bool small = c->bf16 || blocks128 < 512;
  KernelTimer kt(c, A2P_KERNEL_GEMM);
  int rc;
  // 16-bit launches of at most one 64x128 workgroup per CU (config 0: 480 rows) are K/64 serial memory round trips with the
  // 2-deep ring; they take the 4-deep one (A2P_GEMM_RING2=1 keeps the 2-deep ring for A/B runs)
  static const bool ring2 = getenv("A2P_GEMM_RING2") != nullptr;
  const int64_t blocks64 = (int64_t)((p.N + 127) / 128) * ((p.M + 63) / 64);
  // narrow tap-accumulating launches (the body model's dilated conv tail: N <= 128, i.e. one column tile; 312 workgroups of 64 rows
  // at B=16 = 1.2 per CU, each a chain of 18 k-tile round trips): 32-row tiles double the workgroups per CU (A2P_GEMM_MT1=0|1)
  static const int mt1 = getenv("A2P_GEMM_MT1") ? atoi(getenv("A2P_GEMM_MT1")) : 1;
  static const int ring4_blocks = getenv("A2P_GEMM_RING4_BLOCKS") ? atoi(getenv("A2P_GEMM_RING4_BLOCKS")) : 256;   // (A/B: the 4-deep ring up to this many 64-row workgroups)
  if (c->bf16 && mt1 && p.ntaps > 1 && p.N <= 128 && blocks64 <= 3 * 256) rc = gemm_dispatch<h16_t, 1>(kt, p, s);
  else if (c->bf16 && small && blocks64 <= ring4_blocks && p.ntaps == 1 && !ring2) rc = gemm_dispatch<h16_t, 2, 4>(kt, p, s);
  else if (c->bf16) rc = small ? gemm_dispatch<h16_t, 2>(kt, p, s) : gemm_dispatch<h16_t, 4>(kt, p, s);
  else rc = small ? gemm_dispatch<float, 2>(kt, p, s) : gemm_dispatch<float, 4>(kt, p, s);
  CHK(rc);
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_skinny(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                         int M, int N, int K, int act, hipStream_t s) {
  ARG(K % 64 == 0 && N % 16 == 0, "skinny gemm: bad shape N=%d K=%d", N, K);
  for (int m0 = 0; m0 < M; m0 += 64) {
    SkinnyP p;
    p.A = A + (int64_t)m0 * lda; p.W = W; p.bias = bias; p.out = out + (int64_t)m0 * ldo;
    p.lda = lda; p.ldw = ldw; p.ldo = ldo; p.M = (M - m0 < 64) ? M - m0 : 64; p.N = N; p.K = K; p.act = act;
    skinny_gemm_kernel<<<N / 16, 256, 0, s>>>(p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

struct SkinnyArgs {
  const float* A; int64_t lda; const float* W; int64_t ldw; const float* bias; float* out; int64_t ldo; int M, N, K, act;
};
// three independent skinny GEMMs of at most 64 rows each in one launch (kernels_gemm.h skinny_gemm_group_kernel)
static int launch_skinny3(const SkinnyArgs (&a)[3], hipStream_t s) {
  bool one = true;
  for (const SkinnyArgs& q : a) one = one && q.M <= 64 && q.K % 64 == 0 && q.N % 16 == 0;
  if (!one) {
    for (const SkinnyArgs& q : a) CHK(launch_skinny(q.A, q.lda, q.W, q.ldw, q.bias, q.out, q.ldo, q.M, q.N, q.K, q.act, s));
    return 0;
  }
  SkinnyG3 g;
  for (int i = 0; i < 3; ++i) {
    SkinnyP& p = g.p[i];
    p.A = a[i].A; p.W = a[i].W; p.bias = a[i].bias; p.out = a[i].out; p.lda = a[i].lda; p.ldw = a[i].ldw; p.ldo = a[i].ldo;
    p.M = a[i].M; p.N = a[i].N; p.K = a[i].K; p.act = a[i].act;
  }
  g.nb0 = a[0].N / 16; g.nb1 = g.nb0 + a[1].N / 16;
  skinny_gemm_group_kernel<<<g.nb1 + a[2].N / 16, 256, 0, s>>>(g);
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_ln_rope(a2p_ctx* c, bool as_f32, const float* x, int64_t ldx, const float* gamma, const float* beta,
                          void* out_n, void* out_r, int64_t ldo, int rows, int rows_per_seq, int pos_off, hipStream_t s) {
  LnRopeP p;
  p.x = x; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.cs = reinterpret_cast<const float2*>(c->rope_cs.p);
  p.out_n = out_n; p.out_r = out_r; p.ldo = ldo; p.rows = rows; p.rows_per_seq = rows_per_seq; p.pos_off = pos_off;
  const int grid = (rows + 3) / 4;
  KernelTimer kt(c, A2P_KERNEL_LNROPE);
  const bool b16 = c->bf16 && !as_f32;
  if (c->d == 512) {
    if (b16) A2P_LAUNCH(kt, (ln_rope_kernel<h16_t, 8>), grid, 256, s, p);
    else A2P_LAUNCH(kt, (ln_rope_kernel<float, 8>), grid, 256, s, p);
  } else {
    if (b16) A2P_LAUNCH(kt, (ln_rope_kernel<h16_t, 4>), grid, 256, s, p);
    else A2P_LAUNCH(kt, (ln_rope_kernel<float, 4>), grid, 256, s, p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_attn(a2p_ctx* c, const AttnP& p0, int nseq, int kind, hipStream_t s, bool ksplit = false) {
  AttnP p = p0;
  // small forwards (decoder_layer_small): the waves of a workgroup split the keys instead of the queries (kernels_attn.h)
  if ((ksplit || c->opt.force_ksplit) && c->bf16 && !c->opt.no_ksplit && (c->DH == 64 || c->DH == 32) && p.ldvt % 8 == 0 && p.S_tail <= 2) {
    // waves x query tiles per wave: A2P_KSPLIT_NW / A2P_KSPLIT_QT (experiment switches)
    const int ntiles = (p.S_main + p.S_tail + 63) / 64;
    // measured at 2 x 240 frames (profiles/r03_ksplit_ab.txt): 16 queries per wave beat 32 (the tile arithmetic of a lone wave per
    // SIMD is serial), 8 waves beat 4 once a 4-wave workgroup would walk more than two tiles per wave (800 keys = 13 tiles)
    const bool w8 = c->opt.ksplit_nw == 8 || (c->opt.ksplit_nw == 0 && ntiles > 8);
    const int qt = c->opt.ksplit_qt == 2 ? 2 : 1;
    p.nq = (p.Tq + 16 * qt - 1) / (16 * qt); p.nheads = c->H; p.nseq = nseq; p.xcd_remap = 0;
    p.stat_max = c->nonfinite.p ? reinterpret_cast<int*>(c->nonfinite.p) + 1 : nullptr;   // the envelope check covers the small forwards too
    dim3 grid(p.nq * c->H * nseq);
    KernelTimer kt(c, kind);
#define A2P_KS(DH_) \
    do { \
      if (w8 && qt == 1) A2P_LAUNCH(kt, (attn_ksplit_kernel<DH_, 8, 1>), grid, 512, s, p); \
      else if (w8) A2P_LAUNCH(kt, (attn_ksplit_kernel<DH_, 8, 2>), grid, 512, s, p); \
      else if (qt == 1) A2P_LAUNCH(kt, (attn_ksplit_kernel<DH_, 4, 1>), grid, 256, s, p); \
      else A2P_LAUNCH(kt, (attn_ksplit_kernel<DH_, 4, 2>), grid, 256, s, p); \
    } while (0)
    if (c->DH == 64) A2P_KS(64);
    else A2P_KS(32);
#undef A2P_KS
    HIPCHK(hipGetLastError());
    return 0;
  }
  // A2P_ATTN_WAVES=2 (16-bit modes): 2-wave workgroups of 64 queries -- a perfectly even 5 workgroups per CU at B=8, but every K/V
  // tile then feeds half as many queries: measured 97 vs 70 us for the cross attention (experiment switch, kernels_attn.h NWV)
  p.stat_max = (c->DH != 128 && c->nonfinite.p) ? reinterpret_cast<int*>(c->nonfinite.p) + 1 : nullptr;   // denoiser attentions only
  static const bool two = getenv("A2P_ATTN_WAVES") && atoi(getenv("A2P_ATTN_WAVES")) == 2;
  const int nwv = (c->bf16 && two) ? 2 : 4;
  p.nq = (p.Tq + 32 * nwv - 1) / (32 * nwv); p.nheads = c->H; p.nseq = nseq;
  static const bool no_remap = getenv("A2P_ATTN_NO_REMAP") != nullptr;   // A/B switch
  p.xcd_remap = (!no_remap && (c->H * nseq) % 8 == 0) ? 1 : 0;
  if (c->bf16 && c->opt.attn2 && (c->DH == 64 || c->DH == 32) && p.ldvt % 8 == 0) {   // round 5: balanced 8-wave pipeline kernel
    p.nq = (p.Tq + 319) / 320;
    dim3 grid2(p.nq * c->H * nseq);
    KernelTimer kt2(c, kind);
    if (c->DH == 64) A2P_LAUNCH(kt2, (attn2_kernel<64, 3, 2>), grid2, 512, s, p);
    else A2P_LAUNCH(kt2, (attn2_kernel<32, 3, 2>), grid2, 512, s, p);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // Round 6: attn3_kernel (kernels_attn3.h) -- one wave per SIMD, 80 queries per wave, ISA-level tile body with a lazy softmax
  // reference.  Measured stand-alone against attn_kernel (scratch/attn3_bench.hip, profiles/r06_attn3_bench.txt): x1.35 on the B=8
  // cross attention (2000 keys, 256 workgroups = one per CU), x1.12 on its self attention, x1.07 on the B=32 cross attention, but
  // x0.85 on short key ranges once the launch is several rounds of workgroups (its per-workgroup prologue is longer), and slower on
  // the body model's short key ranges -- hence the rule.
  if (c->bf16 && c->opt.attn3 && !c->opt.attn2 && (c->DH == 64 || c->DH == 32) && p.ldvt % 8 == 0) {
    const int nq3 = (p.Tq + 319) / 320;
    const int64_t wgs = (int64_t)nq3 * c->H * nseq;
    // v5 (row sums on the matrix pipe; profiles/r06_attn3_bench_v5.txt): also x1.19 on the body model's cross attention (head_dim 32, 2000 keys, 512 workgroups:
    // 81 vs 97 us); its self attention (600 keys) and the B=32 face self attention stay x0.9
    const int S3 = p.S_main + p.S_tail;
    // ... and NOT below one attn_kernel workgroup per CU (profiles/r06_attn3_small_batch.txt: up to 6 sequences of 600 frames attn_kernel's 128-query workgroups
    // are one round of <= 240 and finish in 32-36 us (cross) / 13-14 us (self) where attn3's 320-query workgroups take their fixed 41-44 / 19 us: x0.7-0.8;
    // from 8 sequences on attn_kernel needs a second round and attn3 leads x1.18 / x1.02)
    const int64_t wgs1 = (int64_t)p.nq * c->H * nseq;   // attn_kernel's grid
    const bool wins = p.Tq >= 160 && wgs1 > 256 && ((c->DH == 64 && (S3 >= 1024 || wgs <= 256)) || (c->DH == 32 && S3 >= 1024));
    if (c->opt.attn3 >= 2 || wins) {
      p.nq = nq3;
      dim3 grid3((unsigned)wgs);
      KernelTimer kt3(c, kind);
      if (c->DH == 64) A2P_LAUNCH(kt3, (attn3_kernel<64>), grid3, 256, s, p);
      else A2P_LAUNCH(kt3, (attn3_kernel<32>), grid3, 256, s, p);
      HIPCHK(hipGetLastError());
      ++c->attn3_launches;
      return 0;
    }
  }
  dim3 grid(p.nq * c->H * nseq);
  KernelTimer kt(c, kind);
  if (c->DH == 128) {  // lip regressor of the audio front end (4 heads x 128), fp32 only
    ARG(!c->bf16, "head_dim 128 is instantiated for fp32 only");
    A2P_LAUNCH(kt, (attn_kernel<float, 128>), grid, 256, s, p);
  } else if (c->DH == 64) {
    if (c->bf16 && nwv == 2) A2P_LAUNCH(kt, (attn_kernel<h16_t, 64, 0, 2>), grid, 128, s, p);
    else if (c->bf16) A2P_LAUNCH(kt, (attn_kernel<h16_t, 64>), grid, 256, s, p);
    else A2P_LAUNCH(kt, (attn_kernel<float, 64>), grid, 256, s, p);
  } else {
    if (c->bf16 && nwv == 2) A2P_LAUNCH(kt, (attn_kernel<h16_t, 32, 0, 2>), grid, 128, s, p);
    else if (c->bf16) A2P_LAUNCH(kt, (attn_kernel<h16_t, 32>), grid, 256, s, p);
    else A2P_LAUNCH(kt, (attn_kernel<float, 32>), grid, 256, s, p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_cast(a2p_ctx* c, const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int cols, int cols_pad,
                       const uint8_t* keep, hipStream_t s, int src_col_stride = 1) {
  const int64_t n = rows * cols_pad;
  const int grid = (int)((n + 255) / 256);
  if (c->bf16) cast_pad_kernel<h16_t><<<grid, 256, 0, s>>>(src, lds, src_col_stride, (h16_t*)dst, ldd, rows, cols, cols_pad, keep);
  else cast_pad_kernel<float><<<grid, 256, 0, s>>>(src, lds, src_col_stride, (float*)dst, ldd, rows, cols, cols_pad, keep);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static const char* kFilmNames[4] = {"film1", "film2", "film3", "film2a"};

static bool ignorable_weight(const std::string& n) {
  auto starts = [&](const char* p) { return n.rfind(p, 0) == 0; };
  if (starts("audio_model.") || starts("lip_model.") || starts("transformer.") || starts("tokenizer.")) return true;
  const std::string suf = "rotary.freqs";
  if (n.size() >= suf.size() && n.compare(n.size() - suf.size(), suf.size(), suf) == 0) return true;
  return false;
}

static void expected_weights(const a2p_ctx* c, std::map<std::string, int64_t>& e) {
  const int64_t d = c->d, C = c->C, ff = c->ff;
  e["null_cond_embed"] = (int64_t)c->cfg.emb_len * d;
  e["null_cond_hidden"] = d;
  e["time_mlp.1.weight"] = 4 * d * d; e["time_mlp.1.bias"] = 4 * d;
  e["to_time_cond.0.weight"] = d * 4 * d; e["to_time_cond.0.bias"] = d;
  e["to_time_tokens.0.weight"] = 2 * d * 4 * d; e["to_time_tokens.0.bias"] = 2 * d;
  e["norm_cond.weight"] = d; e["norm_cond.bias"] = d;
  e["input_projection.weight"] = d * C; e["input_projection.bias"] = d;
  e["cond_projection.weight"] = d * c->Fc; e["cond_projection.bias"] = d;
  e["non_attn_cond_projection.0.weight"] = d; e["non_attn_cond_projection.0.bias"] = d;
  e["non_attn_cond_projection.1.weight"] = d * d; e["non_attn_cond_projection.1.bias"] = d;
  e["non_attn_cond_projection.3.weight"] = d * d; e["non_attn_cond_projection.3.bias"] = d;
  e["final_layer.weight"] = C * d; e["final_layer.bias"] = C;
  auto attn = [&](const std::string& p) {
    e[p + ".in_proj_weight"] = 3 * d * d; e[p + ".in_proj_bias"] = 3 * d;
    e[p + ".out_proj.weight"] = d * d; e[p + ".out_proj.bias"] = d;
  };
  auto norm = [&](const std::string& p) { e[p + ".weight"] = d; e[p + ".bias"] = d; };
  if (c->pose) {
    e["null_pose_embed"] = (int64_t)c->KFmax * d;
    e["frame_cond_projection.weight"] = d * c->Kd; e["frame_cond_projection.bias"] = d;
    norm("frame_norm_cond");
    const int64_t hid = C > 256 ? C : 256;
    const int64_t ci[6] = {C, hid, C, C, C, C}, co[6] = {hid, C, C, C, C, C};
    for (int i = 0; i < 6; ++i) {
      e["post_pose_layers." + std::to_string(i) + ".weight"] = co[i] * ci[i] * 3;
      e["post_pose_layers." + std::to_string(i) + ".bias"] = co[i];
    }
    e["final_conv.weight"] = C * C; e["final_conv.bias"] = C;
  } else {
    for (int i = 0; i < 2; ++i) {
      const std::string p = "cond_encoder." + std::to_string(i) + ".";
      attn(p + "self_attn");
      e[p + "linear1.weight"] = ff * d; e[p + "linear1.bias"] = ff;
      e[p + "linear2.weight"] = d * ff; e[p + "linear2.bias"] = d;
      norm(p + "norm1"); norm(p + "norm2");
    }
  }
  for (int l = 0; l < c->L; ++l) {
    const std::string p = "seqTransDecoder.stack." + std::to_string(l) + ".";
    attn(p + "self_attn"); attn(p + "multihead_attn");
    if (c->pose) attn(p + "multihead_attn2");
    e[p + "linear1.weight"] = ff * d; e[p + "linear1.bias"] = ff;
    e[p + "linear2.weight"] = d * ff; e[p + "linear2.bias"] = d;
    norm(p + "norm1"); norm(p + "norm2"); norm(p + "norm3");
    if (c->pose) norm(p + "norm2a");
    for (int f = 0; f < c->F; ++f) {
      e[p + kFilmNames[f] + ".block.1.weight"] = 2 * d * d;
      e[p + kFilmNames[f] + ".block.1.bias"] = 2 * d;
    }
  }
}

static float* W32(a2p_ctx* c, const std::string& n) { return c->w.at(n).f(); }

// split-operand copy [rows][3 * rup(cols, 64)] = [hi | hi | lo] of a [rows, cols] fp32 matrix (row stride lds, column stride cs)
static int make_wt3(a2p_ctx* c, Buf& b, const float* src, int64_t lds, int cs, int rows, int cols, hipStream_t s) {
  const int kp = rup(cols, 64);
  CHK(buf_alloc(b, (size_t)rows * 3 * kp * 2));
  const int64_t n = (int64_t)rows * kp;
  split3_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(src, lds, cs, reinterpret_cast<h16_t*>(b.p), rows, cols, kp, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

// compute-dtype copy of a [rows, cols] fp32 matrix, K padded to a multiple of 64
static int make_wt(a2p_ctx* c, const std::string& name, const float* src, int rows, int cols, hipStream_t s, Buf* into = nullptr) {
  const int kp = rup(cols, 64);
  Buf& b = into ? *into : c->wt[name];
  CHK(buf_alloc(b, (size_t)rows * kp * c->esz));
  return launch_cast(c, src, cols, b.p, kp, rows, cols, kp, nullptr, s);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" const char* a2p_last_error(void) { return g_err; }
#ifdef A2P_HALF
extern "C" const char* a2p_version(void) { return "a2p_hip 0.2 (gfx950, 16-bit operands = IEEE half)"; }
#else
extern "C" const char* a2p_version(void) { return "a2p_hip 0.2 (gfx950, 16-bit operands = bfloat16)"; }
#endif

extern "C" int a2p_ctx_create(const a2p_config* cfg, a2p_ctx** out) {
  ARG(cfg && out, "null argument");
  ARG(cfg->latent_dim == 256 || cfg->latent_dim == 512, "latent_dim must be 256 or 512 (got %d)", cfg->latent_dim);
  ARG(cfg->num_heads > 0 && cfg->latent_dim % cfg->num_heads == 0, "bad num_heads");
  const int dh = cfg->latent_dim / cfg->num_heads;
  ARG(dh == 32 || dh == 64, "head_dim must be 32 or 64 (got %d)", dh);
  ARG(cfg->ff_size % 64 == 0 && cfg->nfeats % 4 == 0 && cfg->nfeats <= 256, "bad ff_size/nfeats");
  ARG(cfg->max_batch >= 1 && cfg->max_frames >= 16 && cfg->emb_len >= 16 && cfg->num_layers >= 1, "bad capacity");
  ARG(cfg->precision == A2P_PREC_F32 || cfg->precision == A2P_PREC_BF16, "bad precision");
  a2p_ctx* c = new a2p_ctx();
  c->cfg = *cfg;
  load_opts(c->opt);
  c->use_arena = !getenv("A2P_NO_ARENA");
  ArenaScope scope(c->use_arena ? &c->arena : nullptr);
  c->d = cfg->latent_dim; c->H = cfg->num_heads; c->DH = dh; c->L = cfg->num_layers; c->C = cfg->nfeats;
  c->Cpad = rup(cfg->nfeats, 64); c->ff = cfg->ff_size; c->pose = cfg->data_format == A2P_POSE; c->F = c->pose ? 4 : 3;
  c->Fc = cfg->cond_feature_dim; c->FcPad = rup(c->Fc, 64); c->Kd = cfg->keyframe_dim; c->KdPad = rup(c->Kd, 64);
  c->Tmax = cfg->max_frames; c->S0max = cfg->emb_len; c->Sld = rup(cfg->emb_len + 2, 64);
  c->Tld = rup(c->Tmax, 64) > c->Sld ? rup(c->Tmax, 64) : c->Sld;
  c->Bmax = cfg->max_batch; c->Nmax = 2 * c->Bmax;
  c->KFmax = (c->Tmax + cfg->keyframe_step - 1) / cfg->keyframe_step;
  c->bf16 = cfg->precision == A2P_PREC_BF16; c->esz = c->bf16 ? 2 : 4;
  c->tail32 = c->bf16 && !getenv("A2P_TAIL16");
  c->tail_x3 = c->tail32 && !getenv("A2P_TAIL_F32");
  const size_t tsz = c->tail_x3 ? 6 : (c->tail32 ? 4 : c->esz);   // bytes per element of the exact islands' buffers (split rows: 3 x 16 bit)
  ARG(c->KFmax <= 64, "too many keyframes");
  const int64_t r1 = (int64_t)c->Nmax * c->Tmax, r2 = (int64_t)c->Bmax * c->S0max;
  c->rows_cap = (r1 > r2 ? r1 : r2) + 128;
  c->conv_rows = (int64_t)c->Nmax * (c->Tmax + 24) + 64;
  const int64_t R = c->rows_cap, d = c->d;
  int rc = 0;
  auto A = [&](Buf& b, size_t bytes) { if (rc == 0) rc = buf_alloc(b, bytes); };
  A(c->x, R * d * 4); A(c->xn, R * d * c->esz); A(c->xr, R * d * c->esz); A(c->qk, R * 2 * d * c->esz);
  A(c->ao, R * d * c->esz); A(c->hff, R * c->ff * c->esz);
  A(c->vt, (size_t)c->Nmax * d * c->Tld * c->esz);
  A(c->inpack, (size_t)c->Bmax * c->Tmax * c->Cpad * tsz);
  A(c->mo, (size_t)c->conv_rows * c->C * 4);
  if (c->tail_x3) A(c->t3, (size_t)R * 3 * d * 2);   // split rows of the residual stream for final_layer
  if (c->pose) {
    A(c->cb[0], (size_t)c->conv_rows * 128 * tsz); A(c->cb[1], (size_t)c->conv_rows * 256 * tsz);
    A(c->cb[2], (size_t)c->conv_rows * 128 * tsz); A(c->cb[3], (size_t)c->conv_rows * 128 * tsz);
  }
  const int64_t B = c->Bmax, N = c->Nmax, LF = (int64_t)c->L * c->F;
  A(c->emb, B * d * 4); A(c->th, B * 4 * d * 4); A(c->tct, B * 3 * d * 4); A(c->tvec, N * d * 4); A(c->mt, N * d * 4);
  A(c->tokn, B * 2 * d * 4); A(c->tokr, B * 2 * d * 4); A(c->film, N * LF * 2 * d * 4);
  A(c->ktail, B * 2 * c->L * d * 4); A(c->vtail, B * 2 * c->L * d * 4);
  A(c->ce_pack, (size_t)B * c->S0max * c->FcPad * c->esz);
  A(c->pooled, B * d * 4); A(c->tmpa, B * d * 4); A(c->tmpb, B * d * 4);
  A(c->hidden, (B + 1) * d * 4);
  A(c->kc, (size_t)(B + 1) * c->Sld * c->L * d * c->esz); A(c->vtc, (size_t)(B + 1) * c->L * d * c->Sld * c->esz);
  if (c->pose) {
    A(c->k2c, (size_t)(B + 1) * 64 * c->L * d * c->esz); A(c->vt2c, (size_t)(B + 1) * c->L * d * 64 * c->esz);
    A(c->kf_pack, (size_t)B * c->KFmax * c->KdPad * c->esz); A(c->kf_tok, (size_t)B * c->KFmax * d * 4);
  }
  A(c->slot_cond, B * 4); A(c->slot_unc, B * 4); A(c->slot_cfg, N * 4); A(c->nonfinite, 64);
  // logit maximum: "none yet" (0x80808080 < every ordered float).  Synchronous: callers run on non-blocking streams, which do not
  // order behind the null stream -- a first attention could otherwise atomicMax before the sentinel lands and lose its value
  if (rc == 0 && (hipMemsetAsync(reinterpret_cast<int*>(c->nonfinite.p) + 1, 0x80, 4, nullptr) != hipSuccess ||
                  hipStreamSynchronize(nullptr) != hipSuccess)) rc = A2P_ERR_HIP;
  if (getenv("A2P_CHAIN_CLK")) A(c->clk, 64 * 32 * 8 + 2 * 64 * 8);   // + phase stamps of the diagnostic build (-DA2P_STAMPS)
  if (rc == 0 && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) rc = A2P_ERR_HIP;
  for (int i = 0; i < 8 && rc == 0; ++i)
    if (hipEventCreateWithFlags(&c->ev_fork_pool[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join_pool[i], hipEventDisableTiming) != hipSuccess)
      rc = A2P_ERR_HIP;
  if (rc == A2P_ERR_HIP) set_err("side stream / event creation failed");
  if (rc != 0) {
    a2p_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

extern "C" int a2p_set_batch_hint(a2p_ctx* c, int32_t global_batch) {
  ARG(c && global_batch >= 0, "bad arguments");
  c->batch_hint = global_batch;
  return 0;
}

static void graphs_drop(a2p_ctx* c);   // a2p_lib_run.h: waits for the device, then destroys every captured forward

extern "C" int a2p_reload_env(a2p_ctx* c) {
  ARG(c, "null ctx");
  load_opts(c->opt);
  ++c->graph_epoch;   // captured forwards carry the kernel choices of the switches they were captured under:
  graphs_drop(c);     // evicted here, not left to pile up behind a key no forward will ask for again
  return 0;
}

extern "C" int a2p_ctx_destroy(a2p_ctx* c) {
  if (!c) return 0;
  hipDeviceSynchronize();
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
  c->graphs.clear();
  if (c->gstream) (void)hipStreamDestroy(c->gstream);
  for (auto& kv : c->w) buf_free(kv.second);
  for (auto& kv : c->wt) buf_free(kv.second);
  Buf* all[] = {&c->rope_cs, &c->rope_cst, &c->time_freq, &c->film_w, &c->film_b, &c->tct_w, &c->tct_b, &c->tct_table, &c->cak_w32, &c->cak_b, &c->cav_w32,
                &c->cav_b, &c->cak_wt, &c->cav_wt, &c->ca2k_wt, &c->ca2v_wt, &c->ca2k_b, &c->ca2v_b, &c->hidden, &c->kc, &c->vtc,
                &c->k2c, &c->vt2c, &c->slot_cond, &c->slot_unc, &c->slot_cfg, &c->x, &c->xn, &c->xr, &c->qk, &c->vt, &c->ao,
                &c->hff, &c->inpack, &c->mo, &c->cb[0], &c->cb[1], &c->cb[2], &c->cb[3], &c->emb, &c->th, &c->tct, &c->tvec,
                &c->mt, &c->tokn, &c->tokr, &c->film, &c->ktail, &c->vtail, &c->ce_pack, &c->pooled, &c->tmpa, &c->tmpb,
                &c->kf_pack, &c->kf_tok, &c->clk, &c->t3, &c->nonfinite, &c->tail_w, &c->tail_b};
  for (Buf* b : all) buf_free(*b);
  for (int i = 0; i < 7; ++i) buf_free(c->conv_wt[i]);
  for (auto& b : c->ch_stream) buf_free(b);
  for (auto& b : c->ch_stream4) buf_free(b);
  for (auto& b : c->ch_stream4w) buf_free(b);
  buf_free(c->ch_stream4_in); buf_free(c->ch_aux_in);
  for (auto* m : {&c->ch_tune, &c->ch_tune4})
    for (auto& kv : *m)
      for (auto& sm : kv.second.samples) { (void)hipEventDestroy(std::get<1>(sm)); (void)hipEventDestroy(std::get<2>(sm)); }
  for (auto& b : c->ch_aux) buf_free(b);
  for (void* slab : c->arena.slabs) (void)hipFree(slab);
  for (int i = 0; i < 8; ++i) {
    if (c->ev_fork_pool[i]) (void)hipEventDestroy(c->ev_fork_pool[i]);
    if (c->ev_join_pool[i]) (void)hipEventDestroy(c->ev_join_pool[i]);
  }
  if (c->side) (void)hipStreamDestroy(c->side);
  for (auto& e : c->evs) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  delete c;
  return 0;
}

extern "C" int a2p_set_weight(a2p_ctx* c, const char* name, const float* data, int64_t numel, void* stream) {
  ARG(c && name && data, "null argument");
  const std::string n(name);
  if (ignorable_weight(n)) return 1;
  std::map<std::string, int64_t> e;
  expected_weights(c, e);
  auto it = e.find(n);
  if (it == e.end()) {
    set_err("unexpected parameter '%s' (reference: load_model asserts no unexpected keys)", name);
    return A2P_ERR_NOWEIGHT;
  }
  if (it->second != numel) {
    set_err("parameter '%s': expected %lld elements, got %lld", name, (long long)it->second, (long long)numel);
    return A2P_ERR_NOWEIGHT;
  }
  Buf& b = c->w[n];
  ArenaScope scope(c->use_arena && !b.p ? &c->arena : nullptr);  // re-uploads reuse the slot (same size by contract)
  if (!b.p) CHK(buf_alloc(b, (size_t)numel * 4));
  HIPCHK(hipMemcpyAsync(b.p, data, (size_t)numel * 4, hipMemcpyDefault, (hipStream_t)stream));
  c->finalized = false;
  c->prepared = false;
  return 0;
}

// K / V^T of `rows` memory tokens (already normalised+rotated in xr / normalised in xn) for all layers
static int project_kv_all(a2p_ctx* c, const void* xr, const void* xn, int rows, int rows_per_seq, const Buf& kw, const float* kb,
                          const Buf& vw, const float* vb, void* kdst, int kslot_rows, void* vtdst, int first_slot, hipStream_t s) {
  const int d = c->d, LD = c->L * d;
  GemmP pk = gemm_base(xr, d, kw.p, d, kb, kdst, LD, rows, LD, d);
  pk.rows_per_seq = rows_per_seq;
  // row remap: (seq, s) -> (first_slot + seq) * kslot_rows + s     (handled through out pointer + seq pad)
  pk.out = c->offT(kdst, (int64_t)first_slot * kslot_rows * LD);
  pk.epi = EPI_STORE;
  pk.out_seq_pad = kslot_rows - rows_per_seq;
  CHK(launch_gemm(c, pk, s));
  GemmP pv = gemm_base(xn, d, vw.p, d, vb, vtdst, kslot_rows, rows, LD, d);
  pv.epi = EPI_STORE_T;
  pv.rows_per_seq = rows_per_seq;
  pv.t_seq_stride = (int64_t)LD * kslot_rows;
  pv.out = c->offT(vtdst, (int64_t)first_slot * LD * kslot_rows);
  CHK(launch_gemm(c, pv, s));
  return 0;
}

static int chain_build_streams(a2p_ctx* c, hipStream_t s);

extern "C" int a2p_finalize_weights(a2p_ctx* c, void* stream) {
  ARG(c, "null ctx");
  ++c->graph_epoch;   // the compute-dtype weight copies are rebuilt: captured forwards hold the old pointers
  graphs_drop(c);
  ArenaScope scope(c->use_arena ? &c->arena : nullptr);
  hipStream_t s = (hipStream_t)stream;
  std::map<std::string, int64_t> e;
  expected_weights(c, e);
  for (auto& kv : e)
    if (!c->w.count(kv.first)) {
      set_err("missing parameter '%s'", kv.first.c_str());
      return A2P_ERR_NOWEIGHT;
    }
  const int d = c->d, L = c->L, F = c->F;
  // rotary table: freqs_i = 1 / 10000^(2i/d) in fp32 exactly like the reference buffer
  {
    std::vector<float> fr(d / 2);
    for (int i = 0; i < d / 2; ++i) fr[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)d);
    Buf tmp;
    CHK(buf_alloc_tmp(tmp, fr.size() * 4));
    HIPCHK(hipMemcpy(tmp.p, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    const int npos = c->Sld > c->Tld ? c->Sld : c->Tld;
    CHK(buf_alloc(c->rope_cs, (size_t)npos * (d / 2) * 8));
    rope_table_kernel<<<(npos * (d / 2) + 255) / 256, 256, 0, s>>>(tmp.f(), (float2*)c->rope_cs.p, npos, d / 2);
    CHK(buf_alloc(c->rope_cst, (size_t)npos * (d / 2) * 8));   // chain-kernel layout [d/4][npos] x 16 bytes
    rope_table_t_kernel<<<(npos * (d / 4) + 255) / 256, 256, 0, s>>>((const float2*)c->rope_cs.p, (float4*)c->rope_cst.p, npos, d / 2);
    c->rope_npos = npos;
    HIPCHK(hipStreamSynchronize(s));
    buf_free(tmp);
    // SinusoidalPosEmb frequencies (model/utils.py:73-75): exp(arange(half) * -(ln 1e4 / (half-1))), fp32
    const int half = d / 2;
    std::vector<float> tf(half);
    const float em = (float)(-(log(10000.0) / (half - 1)));
    for (int k = 0; k < half; ++k) tf[k] = expf((float)k * em);
    CHK(buf_alloc(c->time_freq, half * 4));
    HIPCHK(hipMemcpy(c->time_freq.p, tf.data(), half * 4, hipMemcpyHostToDevice));
  }
  // compute-dtype copies of every GEMM weight
  if (c->tail_x3) {   // split weight rows [hi | hi | lo]
    CHK(make_wt3(c, c->wt["input_projection.weight"], W32(c, "input_projection.weight"), c->C, 1, d, c->C, s));
    CHK(make_wt3(c, c->wt["final_layer.weight"], W32(c, "final_layer.weight"), d, 1, c->C, d, s));
  } else {
    Fp32Scope f32(c, c->tail32);
    CHK(make_wt(c, "input_projection.weight", W32(c, "input_projection.weight"), d, c->C, s));
    CHK(make_wt(c, "final_layer.weight", W32(c, "final_layer.weight"), c->C, d, s));
  }
  CHK(make_wt(c, "cond_projection.weight", W32(c, "cond_projection.weight"), d, c->Fc, s));
  if (c->pose) CHK(make_wt(c, "frame_cond_projection.weight", W32(c, "frame_cond_projection.weight"), d, c->Kd, s));
  auto attn_wt = [&](const std::string& p) -> int {
    CHK(make_wt(c, p + ".in_proj_weight", W32(c, p + ".in_proj_weight"), 3 * d, d, s));
    CHK(make_wt(c, p + ".out_proj.weight", W32(c, p + ".out_proj.weight"), d, d, s));
    return 0;
  };
  if (!c->pose)
    for (int i = 0; i < 2; ++i) {
      const std::string p = "cond_encoder." + std::to_string(i) + ".";
      CHK(attn_wt(p + "self_attn"));
      CHK(make_wt(c, p + "linear1.weight", W32(c, p + "linear1.weight"), c->ff, d, s));
      CHK(make_wt(c, p + "linear2.weight", W32(c, p + "linear2.weight"), d, c->ff, s));
    }
  // stacked per-layer tensors
  CHK(buf_alloc(c->film_w, (size_t)L * F * 2 * d * d * 4)); CHK(buf_alloc(c->film_b, (size_t)L * F * 2 * d * 4));
  CHK(buf_alloc(c->cak_w32, (size_t)L * d * d * 4)); CHK(buf_alloc(c->cav_w32, (size_t)L * d * d * 4));
  CHK(buf_alloc(c->cak_b, (size_t)L * d * 4)); CHK(buf_alloc(c->cav_b, (size_t)L * d * 4));
  Buf ca2k32, ca2v32;
  if (c->pose) {
    CHK(buf_alloc_tmp(ca2k32, (size_t)L * d * d * 4)); CHK(buf_alloc_tmp(ca2v32, (size_t)L * d * d * 4));
    CHK(buf_alloc(c->ca2k_b, (size_t)L * d * 4)); CHK(buf_alloc(c->ca2v_b, (size_t)L * d * 4));
  }
  auto d2d = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); };
  for (int l = 0; l < L; ++l) {
    const std::string p = "seqTransDecoder.stack." + std::to_string(l) + ".";
    CHK(attn_wt(p + "self_attn"));
    CHK(attn_wt(p + "multihead_attn"));
    if (c->pose) CHK(attn_wt(p + "multihead_attn2"));
    CHK(make_wt(c, p + "linear1.weight", W32(c, p + "linear1.weight"), c->ff, d, s));
    CHK(make_wt(c, p + "linear2.weight", W32(c, p + "linear2.weight"), d, c->ff, s));
    for (int f = 0; f < F; ++f) {
      HIPCHK(d2d(c->film_w.f() + ((size_t)(l * F + f) * 2 * d) * d, W32(c, p + kFilmNames[f] + ".block.1.weight"), (size_t)2 * d * d * 4));
      HIPCHK(d2d(c->film_b.f() + (size_t)(l * F + f) * 2 * d, W32(c, p + kFilmNames[f] + ".block.1.bias"), (size_t)2 * d * 4));
    }
    const float* iw = W32(c, p + "multihead_attn.in_proj_weight");
    const float* ib = W32(c, p + "multihead_attn.in_proj_bias");
    HIPCHK(d2d(c->cak_w32.f() + (size_t)l * d * d, iw + (size_t)d * d, (size_t)d * d * 4));
    HIPCHK(d2d(c->cav_w32.f() + (size_t)l * d * d, iw + (size_t)2 * d * d, (size_t)d * d * 4));
    HIPCHK(d2d(c->cak_b.f() + (size_t)l * d, ib + d, (size_t)d * 4));
    HIPCHK(d2d(c->cav_b.f() + (size_t)l * d, ib + 2 * d, (size_t)d * 4));
    if (c->pose) {
      const float* iw2 = W32(c, p + "multihead_attn2.in_proj_weight");
      const float* ib2 = W32(c, p + "multihead_attn2.in_proj_bias");
      HIPCHK(d2d(ca2k32.f() + (size_t)l * d * d, iw2 + (size_t)d * d, (size_t)d * d * 4));
      HIPCHK(d2d(ca2v32.f() + (size_t)l * d * d, iw2 + (size_t)2 * d * d, (size_t)d * d * 4));
      HIPCHK(d2d(c->ca2k_b.f() + (size_t)l * d, ib2 + d, (size_t)d * 4));
      HIPCHK(d2d(c->ca2v_b.f() + (size_t)l * d, ib2 + 2 * d, (size_t)d * 4));
    }
  }
  CHK(make_wt(c, "", c->cak_w32.f(), L * d, d, s, &c->cak_wt));
  CHK(make_wt(c, "", c->cav_w32.f(), L * d, d, s, &c->cav_wt));
  if (c->pose) {
    CHK(make_wt(c, "", ca2k32.f(), L * d, d, s, &c->ca2k_wt));
    CHK(make_wt(c, "", ca2v32.f(), L * d, d, s, &c->ca2v_wt));
  }
  // time path: [to_time_cond ; to_time_tokens] stacked -> [3d, 4d]
  CHK(buf_alloc(c->tct_w, (size_t)3 * d * 4 * d * 4)); CHK(buf_alloc(c->tct_b, (size_t)3 * d * 4));
  HIPCHK(d2d(c->tct_w.f(), W32(c, "to_time_cond.0.weight"), (size_t)d * 4 * d * 4));
  HIPCHK(d2d(c->tct_w.f() + (size_t)d * 4 * d, W32(c, "to_time_tokens.0.weight"), (size_t)2 * d * 4 * d * 4));
  HIPCHK(d2d(c->tct_b.f(), W32(c, "to_time_cond.0.bias"), (size_t)d * 4));
  HIPCHK(d2d(c->tct_b.f() + d, W32(c, "to_time_tokens.0.bias"), (size_t)2 * d * 4));
  // ... and its outputs for every timestep value: three latency-bound launches of every forward become a row lookup in tpath_post_kernel
  c->tct_rows = 0;
  if (c->opt.time_table > 0) {
    const int n = c->opt.time_table;
    std::vector<int64_t> tv(n);
    for (int i = 0; i < n; ++i) tv[i] = i;
    Buf tt, te, th;
    CHK(buf_alloc_tmp(tt, (size_t)n * 8)); CHK(buf_alloc_tmp(te, (size_t)n * d * 4)); CHK(buf_alloc_tmp(th, (size_t)n * 4 * d * 4));
    HIPCHK(hipMemcpyAsync(tt.p, tv.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    time_embed_kernel<<<(n * (d / 2) + 255) / 256, 256, 0, s>>>(reinterpret_cast<const int64_t*>(tt.p), c->time_freq.f(), te.f(), n, d / 2);
    CHK(launch_skinny(te.f(), d, W32(c, "time_mlp.1.weight"), d, W32(c, "time_mlp.1.bias"), th.f(), 4 * d, n, 4 * d, d, ACT_MISH, s));
    CHK(buf_alloc(c->tct_table, (size_t)n * 3 * d * 4));
    CHK(launch_skinny(th.f(), 4 * d, c->tct_w.f(), 4 * d, c->tct_b.f(), c->tct_table.f(), 3 * d, n, 3 * d, 4 * d, ACT_NONE, s));
    HIPCHK(hipStreamSynchronize(s));   // (tv is host memory; the temporaries go back)
    buf_free(tt); buf_free(te); buf_free(th);
    c->tct_rows = n;
  }
  // slot 0 of the caches: the unconditional branch is batch- and input-invariant
  HIPCHK(d2d(c->hidden.p, W32(c, "null_cond_hidden"), (size_t)d * 4));
  {
    const int rows = c->S0max;
    CHK(launch_ln_rope(c, false, W32(c, "null_cond_embed"), d, W32(c, "norm_cond.weight"), W32(c, "norm_cond.bias"), c->xn.p,
                       c->xr.p, d, rows, rows, 0, s));
    CHK(project_kv_all(c, c->xr.p, c->xn.p, rows, rows, c->cak_wt, c->cak_b.f(), c->cav_wt, c->cav_b.f(), c->kc.p, c->Sld,
                       c->vtc.p, 0, s));
  }
  if (c->pose) {
    const int rows = c->KFmax;
    // null_pose_embed replaces the *normalised* keyframe tokens (model/diffusion.py:331-335): no LayerNorm here
    CHK(launch_ln_rope(c, false, W32(c, "null_pose_embed"), d, nullptr, nullptr, c->xn.p, c->xr.p, d, rows, rows, 0, s));
    CHK(project_kv_all(c, c->xr.p, c->xn.p, rows, rows, c->ca2k_wt, c->ca2k_b.f(), c->ca2v_wt, c->ca2v_b.f(), c->k2c.p, 64,
                       c->vt2c.p, 0, s));
    // conv tail weights [Co, Ci, 3] -> [tap][Co][CiPad]
    const int C = c->C, hid = C > 256 ? C : 256;
    const int ci[7] = {C, hid, C, C, C, C, C}, co[7] = {hid, C, C, C, C, C, C};
    Fp32Scope f32(c, c->tail32 && !c->tail_x3);
    for (int i = 0; i < 7; ++i) {
      const int taps = i < 6 ? 3 : 1, cip = rup(ci[i], 64);
      const std::string nm = i < 6 ? "post_pose_layers." + std::to_string(i) + ".weight" : "final_conv.weight";
      if (c->tail_x3) {   // [tap][Co][3 * CiPad] split rows
        CHK(buf_alloc(c->conv_wt[i], (size_t)taps * co[i] * 3 * cip * 2));
        for (int t = 0; t < taps; ++t) {
          const int64_t n = (int64_t)co[i] * cip;
          split3_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(W32(c, nm) + t, (int64_t)ci[i] * taps, taps,
                                                              reinterpret_cast<h16_t*>(c->conv_wt[i].p) + (int64_t)t * co[i] * 3 * cip, co[i], ci[i], cip, 1);
        }
        HIPCHK(hipGetLastError());
        continue;
      }
      CHK(buf_alloc(c->conv_wt[i], (size_t)taps * co[i] * cip * c->esz));
      for (int t = 0; t < taps; ++t)  // src element (co, ci, tap) at (co*Ci + ci)*taps + tap
        CHK(launch_cast(c, W32(c, nm) + t, (int64_t)ci[i] * taps, c->offT(c->conv_wt[i], (int64_t)t * co[i] * cip), cip, co[i], ci[i],
                        cip, nullptr, s, taps));
    }
  }
  if (c->pose && c->tail_x3 && c->d == 256 && c->C == 104 && !getenv("A2P_NO_FUSED_TAIL")) {
    // fused output tail (kernels_tail.h): final_layer, post_pose_layers.0..5, final_conv as (hi, lo) MFMA operands in consumption order
    struct L { std::string w, b; int Co, Ci, taps, cpin, nt; };
    std::vector<L> ls = {{"final_layer.weight", "final_layer.bias", c->C, c->d, 1, 256, 8}};
    for (int i = 0; i < 6; ++i)
      ls.push_back({"post_pose_layers." + std::to_string(i) + ".weight", "post_pose_layers." + std::to_string(i) + ".bias", i == 0 ? 256 : c->C,
                    i == 1 ? 256 : c->C, 3, i == 1 ? 256 : 128, i == 0 ? 16 : 8});
    ls.push_back({"final_conv.weight", "final_conv.bias", c->C, c->C, 1, 128, 8});
    int64_t total = 0;
    for (size_t i = 0; i < ls.size(); ++i) {
      c->tail_woff[i] = total;
      total += (int64_t)(ls[i].taps * ls[i].cpin / 32) * ls[i].nt * 2 * 512;
    }
    CHK(buf_alloc(c->tail_w, (size_t)total * 2));
    CHK(buf_alloc(c->tail_b, (size_t)8 * 256 * 4));   // zero-filled: the padded output columns carry a zero bias
    for (size_t i = 0; i < ls.size(); ++i) {
      const int kcs = ls[i].taps * ls[i].cpin / 32;
      const int64_t n = (int64_t)kcs * ls[i].nt * 64;
      tail_pack_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(W32(c, ls[i].w), ls[i].Co, ls[i].Ci, ls[i].taps, ls[i].cpin, ls[i].nt, kcs,
                                                            reinterpret_cast<h16_t*>(c->tail_w.p) + c->tail_woff[i]);
      HIPCHK(hipMemcpyAsync(c->tail_b.f() + i * 256, W32(c, ls[i].b), (size_t)ls[i].Co * 4, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipGetLastError());
    c->tail_fused = true;
  }
  HIPCHK(hipStreamSynchronize(s));
  buf_free(ca2k32);
  buf_free(ca2v32);
  if (c->bf16 && (c->d == 512 || c->d == 256) && c->ff == 1024) CHK(chain_build_streams(c, s));
  c->finalized = true;
  return 0;
}

#include "a2p_lib_run.h"
#include "a2p_guide.h"
#include "a2p_frontend.h"
