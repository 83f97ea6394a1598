// Fused attention softmax(scale * q k^T) v with bf16 OPERANDS on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), f32
// accumulation and an f32 online softmax -- the 16-bit form the reference trains its Perceiver in
// (experiments/003_perceiver_processes_single_sat_image_then_rnn.py:40,288-294: Lightning precision=16 runs the two
// einsums of perceiver_pytorch's Attention.forward in half precision).  Tensors stay f32 in memory (the surrounding
// GEMMs / LayerNorms are the exact-f32 kernels); q, k, v, dO are rounded to bf16 (nearest even) on their way into
// registers / LDS, the probabilities and dS when they become matrix operands.
//
// Same decomposition as attention_f32.hip (4 waves x 32 queries, key tiles of 32, S^T layout so that a lane owns ONE
// query's scores), but a contraction of 64 is 4 instructions of 8 passes instead of 32 instructions of 16 passes: 16x
// less matrix-pipe time, which makes staging and the softmax arithmetic the bound.  Operand layouts of the 32x32x16
// instruction: A lane (row = lane % 32, k = 8 (lane / 32) + 0..7), B lane (k = 8 (lane / 32) + 0..7, column = lane % 32):
// every operand is 8 CONSECUTIVE elements, one ds_read_b128 from a row-major bf16 LDS image.  Where the contraction runs
// over an accumulator's rows (P, dS go from registers straight into the B operand) the other operand is needed TRANSPOSED
// and in the accumulator's row order: contraction slots 8 half + 0..7 of step st are rows 16 st + 4 half + {0..3, 8..11},
// i.e. two runs of four consecutive rows -- exactly what two ds_read_b64_tr_b16 (the transposing LDS read: a 16-lane
// group fetches a 4-row x 16-column block and every lane receives one column) deliver from the SAME row-major image.
// (Round 2 first kept separate transposed images filled by 2-byte scatter stores: PMC showed 84 % of the backward's LDS
// cycles as bank conflicts; the largest single source was the dQ reduction scratch, read 32-way conflicted.)
#include "pv_common.h"

namespace pv {

typedef float v16f_b __attribute__((ext_vector_type(16)));

constexpr int BD = 64;         // head dimension
constexpr int BTJ = 32;        // keys per tile
constexpr int B_KLD = 72;      // bf16 per row of a [rows][64 d] image (144 B: 16-byte aligned rows, banks rotate by 36 words)
constexpr int B_TLD = 40;      // bf16 per row of a [.][32] tile (80 B)

constexpr bool ATT_BWD_PREFETCH = false;   // backward: next tile's K / V in registers while the current one is processed

struct AttnGeomB {
  int n_q, n_k, heads;
  long long q_bs, q_rs, k_bs, k_rs;
  float scale;
};

__device__ __forceinline__ int att_acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ bf16x8 att_pack8(const float (&x)[8]) {
  u32x4 w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = pack_bf16_pair(x[2 * j], x[2 * j + 1]);
  return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ bf16x8 att_load8(const float* p, bool ok) {   // 8 consecutive floats (16-byte aligned) -> bf16x8
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (ok) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return att_pack8(x);
}

// n_splits > 1: blockIdx.x = split * n_qtiles + query tile; a workgroup walks the keys [split * keys_per_split, ...) only
// and leaves its UNNORMALISED accumulator, running maximum and running sum in `part` ([split][b][h][query][64 + 2]);
// attn_fwd_combine merges the splits in split order.  (152 (batch, head) groups of one query tile each leave 40 % of
// the chip idle and one wave per SIMD: with 4 splits two or three workgroups share a CU and overlap each other's softmax.)
// KV = float: K / V f32 in memory, rounded to bf16 on the way into LDS; KV = uint16_t: K / V already bf16 in memory (the
// projection that produced them wrote bf16: half the bytes of this HBM-bound kernel -- 128 queries make 64 flop per byte of
// f32 K / V) -- same values, same results
#ifdef PV_DIAG_STAMPS
__device__ unsigned long long attn_fwd_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif
template <typename KV>
__global__ __launch_bounds__(256) void attn_fwd_bf16(const float* __restrict__ q, const KV* __restrict__ k,
                                                      const KV* __restrict__ v, float* __restrict__ o,
                                                      float* __restrict__ lse, AttnGeomB g, int n_splits, int keys_per_split,
                                                      float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) uint16_t Ks[BTJ * B_KLD];   // [key][d]
  __shared__ __attribute__((aligned(16))) uint16_t Vs[BTJ * B_KLD];   // [key][d]; read transposed (ds_read_b64_tr_b16) for V^T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tr_qi = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4_f;
  typedef __attribute__((ext_vector_type(8))) short s16x8_f;
  auto tr8 = [&](const uint16_t* img, int ld, int row_lo, int row_hi, int colbase) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_f*)(img + (row_lo + tr_qi) * ld + colbase + tr_col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_f*)(img + (row_hi + tr_qi) * ld + colbase + tr_col));
    const s16x8_f v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int n_qt = (g.n_q + 127) / 128;
  const int split = blockIdx.x / n_qt, qt = blockIdx.x - split * n_qt;
  const int key_lo = split * keys_per_split;
  const int key_hi = key_lo + keys_per_split < g.n_k ? key_lo + keys_per_split : g.n_k;
  const int i = qt * 128 + wave * 32 + col;           // this lane's query
  const float* qb = q + b * g.q_bs + h * BD;
  const KV* kb = k + b * g.k_bs + h * BD;
  const KV* vb = v + b * g.k_bs + h * BD;
  constexpr bool KV16 = sizeof(KV) == 2;
  // B operand of S^T = K Q^T: lane (query = col, d = 16 ks + 8 half + 0..7)
  bf16x8 qreg[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qreg[ks] = att_load8(qb + (long long)(i < g.n_q ? i : 0) * g.q_rs + 16 * ks + 8 * half, i < g.n_q);

  v16f_b acc0, acc1;   // O^T rows d = 0..31 / 32..63, column = query
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale_log2e = g.scale * 1.44269504088896340736f;

  // staging: thread t moves 8 floats of K and of V of the 32 x 64 tile: key t / 8, d = 8 (t % 8) ..
  const int st_row = tid >> 3, st_col = (tid & 7) * 8;
  // (Four key tiles in flight per thread instead of one were tried for the bf16-stored K / V: 332 against 314 us, and the
  // f32-stored form went 269 -> 357 us: the loop is not waiting for memory.)
  f32x4 k0, k1, v0, v1;      // (KV16: k0 / v0 hold the 8 bf16 values as they are)
  auto load_tile = [&](int j0) {
    const int j = j0 + st_row;
    const bool ok = j < key_hi;
    const KV* kp = kb + (long long)(ok ? j : 0) * g.k_rs + st_col;
    const KV* vp = vb + (long long)(ok ? j : 0) * g.k_rs + st_col;
    k0 = *reinterpret_cast<const f32x4*>(kp);
    v0 = *reinterpret_cast<const f32x4*>(vp);
    if constexpr (!KV16) { k1 = *reinterpret_cast<const f32x4*>(kp + 4); v1 = *reinterpret_cast<const f32x4*>(vp + 4); }
    // (rows beyond the range read row 0 instead of being zeroed: their scores are masked to -inf below, so their p is 0 and
    // any finite K / V will do.  The select that zeroed them made the bf16-stored form WAIT for the load right here, in the
    // middle of the tile it was meant to run under: 2 800 of its 4 500 cycles per tile, found with stamps)
  };
  load_tile(key_lo);
#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, t0, t1;
#define ATT_PHASE(slot) do { PV_STAMP(t1); dg[slot] += t1 - t0; t0 = t1; } while (0)
  PV_STAMP(t0);
#else
#define ATT_PHASE(slot) do { } while (0)
#endif
  for (int j0 = key_lo; j0 < key_hi; j0 += BTJ) {
    __syncthreads();
    ATT_PHASE(0);
    if constexpr (KV16) {
      *reinterpret_cast<f32x4*>(Ks + st_row * B_KLD + st_col) = k0;
      *reinterpret_cast<f32x4*>(Vs + st_row * B_KLD + st_col) = v0;
    } else {
      const float kx[8] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
      *reinterpret_cast<bf16x8*>(Ks + st_row * B_KLD + st_col) = att_pack8(kx);
      const float vx[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      *reinterpret_cast<bf16x8*>(Vs + st_row * B_KLD + st_col) = att_pack8(vx);
    }
    ATT_PHASE(1);      // (includes the wait for this tile's K / V to arrive from memory)
    __syncthreads();
    ATT_PHASE(2);
    if (j0 + BTJ < key_hi) load_tile(j0 + BTJ);
    // S^T tile: rows = keys, column = this lane's query
    v16f_b s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Ks + col * B_KLD + 16 * ks + 8 * half),
                                                  qreg[ks], s, 0, 0, 0);
    // Online softmax with a LAZY reference: p = exp2(scale log2e s - m_run) where m_run is only moved (and the accumulators
    // rescaled) when a tile's maximum exceeds it by more than 2^LAZY -- p then stays below 2^LAZY, which a bf16 operand and the
    // f32 sums hold without loss, and the 33 multiplications + one exponential of a rescale leave almost every tile (the
    // exact-maximum form rescaled in ~2/3 of the tiles of a 3 000-key range).  One fused multiply-add and one v_exp_f32 per
    // score; keys beyond the range are masked in the (single) ragged tile only.
    constexpr float LAZY = 8.f;
    if (j0 + BTJ > key_hi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = j0 + att_acc_row(r, half) < key_hi ? s[r] : -INFINITY;
    }
    ATT_PHASE(3);
    float m_tile = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
    m_tile = fmaxf(m_tile, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64)) * scale_log2e;      // (scale > 0)
    if (__builtin_amdgcn_ballot_w64(m_tile > m_run + LAZY) != 0) {      // wave-uniform: some query of this wave needs a new reference
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // first tile: exp2(-inf) = 0 on zero accumulators
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] *= alpha, acc1[r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, -m_run));
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    ATT_PHASE(4);
    // O^T += V^T P^T: contraction slot 8 half + t of step st is key att_acc_row(8 st + t, half): P comes straight from
    // the S accumulator, V^T from the permuted transposed tile
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const float px[8] = {s[8 * st], s[8 * st + 1], s[8 * st + 2], s[8 * st + 3], s[8 * st + 4], s[8 * st + 5], s[8 * st + 6],
                           s[8 * st + 7]};
      const bf16x8 pb = att_pack8(px);
      // contraction slots 8 half + 0..7 of step st = keys 16 st + 4 half + {0..3, 8..11} of the tile (accumulator-row order)
      const int k_lo = 16 * st + 4 * half;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Vs, B_KLD, k_lo, k_lo + 8, 0), pb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Vs, B_KLD, k_lo, k_lo + 8, 32), pb, acc1, 0, 0, 0);
    }
    ATT_PHASE(5);
#ifdef PV_DIAG_STAMPS
    dg[7] += 1;
#endif
  }
#undef ATT_PHASE
#ifdef PV_DIAG_STAMPS
  {
    const int wid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
    if (lane == 0 && wid < PV_DIAG_WAVES)
      for (int ii = 0; ii < PV_DIAG_SLOTS; ++ii) attn_fwd_diag[(size_t)wid * PV_DIAG_SLOTS + ii] = dg[ii];
  }
#endif
  if (n_splits > 1) {
    if (i < g.n_q) {
      float* pb = part + ((((long long)split * gridDim.z + b) * g.heads + h) * g.n_q + i) * (BD + 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = att_acc_row(r, half);
        pb[d] = acc0[r];
        pb[32 + d] = acc1[r];
      }
      if (half == 0) pb[BD] = m_run * 0.69314718055994530942f, pb[BD + 1] = l_run;   // maximum back in natural-log units
    }
    return;
  }
  if (i < g.n_q) {
    const float inv = 1.0f / l_run;
    float* ob = o + b * g.q_bs + (long long)i * g.q_rs + h * BD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = att_acc_row(r, half);
      ob[d] = acc0[r] * inv;
      ob[32 + d] = acc1[r] * inv;
    }
    if (half == 0) lse[((long long)b * g.heads + h) * g.n_q + i] = m_run * 0.69314718055994530942f + logf(l_run);
  }
}

// merges the key splits of attn_fwd_bf16: thread = (row (b, h, query), d); splits in index order
__global__ __launch_bounds__(256) void attn_fwd_combine(const float* __restrict__ part, float* __restrict__ o,
                                                         float* __restrict__ lse, AttnGeomB g, int n_splits, long long rows,
                                                         int batch) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int d = threadIdx.x & 63;
  if (row >= rows) return;
  const int i = (int)(row % g.n_q);
  const long long bh = row / g.n_q;
  const int h = (int)(bh % g.heads), b = (int)(bh / g.heads);
  float m = -INFINITY;
  for (int s2 = 0; s2 < n_splits; ++s2) m = fmaxf(m, part[((long long)s2 * rows + row) * (BD + 2) + BD]);
  float l = 0.f, acc = 0.f;
  for (int s2 = 0; s2 < n_splits; ++s2) {
    const float* pb = part + ((long long)s2 * rows + row) * (BD + 2);
    const float w = expf(pb[BD] - m);        // a split without keys in range carries m = -inf, l = 0: weight 0
    l += pb[BD + 1] * w;
    acc += pb[d] * w;
  }
  o[b * g.q_bs + (long long)i * g.q_rs + h * BD + d] = acc / l;
  if (d == 0) lse[row] = m + logf(l);
}

// Backward for n_q <= 128: one workgroup per (b, h, key range).  Q and dO of all (<= 128) queries sit in LDS as bf16, both
// row-major (A operands of S = Q K^T, dP = dO V^T) and transposed with the query index in accumulator order (A operands of
// dV^T += dO^T P, dK^T += Q^T dS).  Wave w walks the key tiles w, w+4, ... on its own (no barriers inside the loop): K and V
// of the tile are B operands in registers, K^T a per-wave LDS tile (A operand of dQ^T += K^T dS^T), dS goes through a
// per-wave LDS tile to be read back along the keys.
// KV: float or uint16_t (bf16 K / V in memory), as attn_fwd_bf16.  DKV: float, or uint16_t = dK / dV STORED as bf16 (rounded
// to nearest even; no accumulate form): for a consumer that rounds them to bf16 anyway (the bf16-operand GEMMs of to_kv's
// backward) -- the 1.27 GB of f32 dK / dV were two thirds of this launch's bytes
template <typename KV, typename DKV = float>
__global__ __launch_bounds__(256) void attn_bwd_bf16(const float* __restrict__ q, const KV* __restrict__ k,
                                                      const KV* __restrict__ v, const float* __restrict__ dout,
                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                      float* __restrict__ dq, DKV* __restrict__ dk, DKV* __restrict__ dv,
                                                      AttnGeomB g, int tiles_per_split, long long dq_ss, int accumulate_dkv) {
  // LDS: Q and dO of all (<= 128) queries, row-major bf16 [query][d] (A operands of S = Q K^T and dP = dO V^T with one
  // ds_read_b128; the TRANSPOSED operands of dV^T += dO^T P and dK^T += Q^T dS come from the same images through
  // ds_read_b64_tr_b16 -- round 2 kept separate transposed images, built with 2-byte scatter stores, 35 KB more LDS and
  // one workgroup per CU); per wave the K tile [key][d] (transposed-read for dQ^T += K^T dS^T) and dS [key][query].
  constexpr int QS_B = 128 * B_KLD * 2, KR_B = BTJ * B_KLD * 2, TW_B = BTJ * B_TLD * 2;
  constexpr int SLD = BD + 1;                                         // dQ reduction scratch pitch (floats)
  constexpr int SMEM_B = 2 * QS_B + 4 * (KR_B + TW_B) > 2 * 4 * 32 * SLD * 4 ? 2 * QS_B + 4 * (KR_B + TW_B) : 2 * 4 * 32 * SLD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_B];
  __shared__ __attribute__((aligned(16))) float Ls[128];
  __shared__ __attribute__((aligned(16))) float Ds[128];
  uint16_t* Qs = reinterpret_cast<uint16_t*>(smem);                       // [query][d]
  uint16_t* Os = reinterpret_cast<uint16_t*>(smem + QS_B);                // dO, same layout
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t* Kr = reinterpret_cast<uint16_t*>(smem + 2 * QS_B + wave * (KR_B + TW_B));            // [key of the tile][d]
  uint16_t* Tw = reinterpret_cast<uint16_t*>(smem + 2 * QS_B + wave * (KR_B + TW_B) + KR_B);     // dS [key][query of tile it]
  // transposed 8-element operand: rows row_lo + 0..3 and row_hi + 0..3 of an [row][column] bf16 image, this lane's column
  // = colbase + lane % 32 (16-lane groups fetch 4 x 16 blocks: lane (qi, pi) supplies row qi, 4-column piece pi)
  const int tr_qi = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr8 = [&](const uint16_t* img, int ld, int row_lo, int row_hi, int colbase) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + (row_lo + tr_qi) * ld + colbase + tr_col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + (row_hi + tr_qi) * ld + colbase + tr_col));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const float* qb = q + b * g.q_bs + h * BD;
  const float* ob = dout + b * g.q_bs + h * BD;
  const KV* kb = k + b * g.k_bs + h * BD;
  const KV* vb = v + b * g.k_bs + h * BD;
  constexpr bool KV16 = sizeof(KV) == 2;
  // the next tile's K / V in registers while the current one is processed: with bf16 in memory they take the 32 registers the
  // f32 form needs for the tile in hand (which has none to spare: 256 + 80 registers, one wave per SIMD)
  constexpr bool PREFETCH = KV16 || ATT_BWD_PREFETCH;
  const int n_tiles = (g.n_k + BTJ - 1) / BTJ;
  const int n_qt = (g.n_q + 31) / 32;
  const float scale_log2e = g.scale * 1.44269504088896340736f;
  const int tile0 = blockIdx.x * tiles_per_split;
  const int tile1 = tile0 + tiles_per_split < n_tiles ? tile0 + tiles_per_split : n_tiles;
  dq += blockIdx.x * dq_ss;     // this key range's partial dQ (summed over the splits afterwards)
  // K, V of a tile as B operands: lane (key = col, d = 16 ks + 8 half + 0..7); the NEXT tile's loads are issued before the
  // current tile's arithmetic (one wave per SIMD walks its tiles alone)
  f32x4 kn[4][2], vn[4][2];
  auto load_kv = [&](int tile) {
    const int jj = tile * BTJ + col;
    const bool ok = tile < tile1 && jj < g.n_k;
    const KV* kp = kb + (long long)(ok ? jj : 0) * g.k_rs + 8 * half;
    const KV* vp = vb + (long long)(ok ? jj : 0) * g.k_rs + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {      // (KV16: kn / vn [ks][0] hold the 8 bf16 values as they are)
      kn[ks][0] = *reinterpret_cast<const f32x4*>(kp + 16 * ks);
      vn[ks][0] = *reinterpret_cast<const f32x4*>(vp + 16 * ks);
      if constexpr (!KV16) { kn[ks][1] = *reinterpret_cast<const f32x4*>(kp + 16 * ks + 4); vn[ks][1] = *reinterpret_cast<const f32x4*>(vp + 16 * ks + 4); }
      // (keys beyond the range read key 0: their P and dS are zeroed below (j_ok), so any finite K / V will do -- a select on
      // the loaded values makes the wave wait for them at once)
    }
  };
  if (PREFETCH) load_kv(tile0 + wave);   // in flight while Q / dO are staged
  // ---- stage Q, dO (zero rows beyond n_q), lse (+inf beyond n_q so that P = 0 there), delta ---------------------------
  for (int idx = tid; idx < 128 * 8; idx += 256) {
    const int i = idx >> 3, c8 = (idx & 7) * 8;
    const bool ok = i < g.n_q;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    if (ok) {
      a0 = *reinterpret_cast<const f32x4*>(qb + (long long)i * g.q_rs + c8); a1 = *reinterpret_cast<const f32x4*>(qb + (long long)i * g.q_rs + c8 + 4);
      b0 = *reinterpret_cast<const f32x4*>(ob + (long long)i * g.q_rs + c8); b1 = *reinterpret_cast<const f32x4*>(ob + (long long)i * g.q_rs + c8 + 4);
    }
    const float qx[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const float ox[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    *reinterpret_cast<bf16x8*>(Qs + i * B_KLD + c8) = att_pack8(qx);
    *reinterpret_cast<bf16x8*>(Os + i * B_KLD + c8) = att_pack8(ox);
  }
  if (tid < 128) {
    const long long li = ((long long)b * g.heads + h) * g.n_q + tid;
    Ls[tid] = tid < g.n_q ? lse[li] * 1.44269504088896340736f : INFINITY;   // in log2 units: P = exp2(scale log2e S - lse log2e)
    Ds[tid] = tid < g.n_q ? delta[li] : 0.f;
  }
  __syncthreads();

  v16f_b dqa[4][2];   // dQ^T partial of this wave: [query tile][d half], rows d, column = query
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqa[it][t][r] = 0.f;
  for (int tile = tile0 + wave; tile < tile1; tile += 4) {
    if (!PREFETCH) load_kv(tile);
    const int j0 = tile * BTJ;
    const int j = j0 + col;                       // this lane's key (B-operand column)
    const bool j_ok = j < g.n_k;
    bf16x8 kreg[4], vreg[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (KV16) {
        kreg[ks] = __builtin_bit_cast(bf16x8, kn[ks][0]);
        vreg[ks] = __builtin_bit_cast(bf16x8, vn[ks][0]);
      } else {
        const float kx[8] = {kn[ks][0][0], kn[ks][0][1], kn[ks][0][2], kn[ks][0][3], kn[ks][1][0], kn[ks][1][1], kn[ks][1][2], kn[ks][1][3]};
        const float vx[8] = {vn[ks][0][0], vn[ks][0][1], vn[ks][0][2], vn[ks][0][3], vn[ks][1][0], vn[ks][1][1], vn[ks][1][2], vn[ks][1][3]};
        kreg[ks] = att_pack8(kx);
        vreg[ks] = att_pack8(vx);
      }
      *reinterpret_cast<bf16x8*>(Kr + col * B_KLD + 16 * ks + 8 * half) = kreg[ks];   // row-major; read transposed for dQ
    }
    if (PREFETCH) load_kv(tile + 4);
    v16f_b dv0, dv1, dk0, dk1;   // dV^T / dK^T of this key tile: rows d, column = key
#pragma unroll
    for (int r = 0; r < 16; ++r) dv0[r] = 0.f, dv1[r] = 0.f, dk0[r] = 0.f, dk1[r] = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it < n_qt) {
        // S[i][j] = Q K^T and dP[i][j] = dO V^T: rows = queries of tile it, column = key
        v16f_b sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              *reinterpret_cast<const bf16x8*>(Qs + (32 * it + col) * B_KLD + 16 * ks + 8 * half), kreg[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              *reinterpret_cast<const bf16x8*>(Os + (32 * it + col) * B_KLD + 16 * ks + 8 * half), vreg[ks], dp, 0, 0, 0);
        }
        // P = exp(scale S - lse_i), dS = scale * P * (dP - delta_i); masked keys give P = 0
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {     // registers 4 q4 + 0..3 hold queries 32 it + 8 q4 + 4 half + 0..3: one 16-byte read each
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + it * 32 + 8 * q4 + 4 * half);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + it * 32 + 8 * q4 + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q4 + e;
            const float pv_ = j_ok ? __builtin_amdgcn_exp2f(sc[r] * scale_log2e - l4[e]) : 0.f;   // one v_exp_f32
            sc[r] = pv_;                                      // sc now holds P
            dp[r] = g.scale * pv_ * (dp[r] - d4[e]);          // dp now holds dS
          }
        }
        // dS of this lane's key, queries 8 g + 4 half + 0..3 per register quad g: four packed 8-byte stores into [key][query]
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          u32x2 w2;
          w2[0] = pack_bf16_pair(dp[4 * q4], dp[4 * q4 + 1]);
          w2[1] = pack_bf16_pair(dp[4 * q4 + 2], dp[4 * q4 + 3]);
          *reinterpret_cast<u32x2*>(Tw + col * B_TLD + 8 * q4 + 4 * half) = w2;
        }
        // dV^T += dO^T P, dK^T += Q^T dS: contraction slot 8 half + t of step st is query att_acc_row(8 st + t, half)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const float px[8] = {sc[8 * st], sc[8 * st + 1], sc[8 * st + 2], sc[8 * st + 3], sc[8 * st + 4], sc[8 * st + 5],
                               sc[8 * st + 6], sc[8 * st + 7]};
          const float sx[8] = {dp[8 * st], dp[8 * st + 1], dp[8 * st + 2], dp[8 * st + 3], dp[8 * st + 4], dp[8 * st + 5],
                               dp[8 * st + 6], dp[8 * st + 7]};
          const bf16x8 pb = att_pack8(px), sb = att_pack8(sx);
          // contraction slots 8 half + 0..7 of step st = queries 32 it + 16 st + 4 half + {0..3, 8..11}: two 4-row runs
          const int q_lo = 32 * it + 16 * st + 4 * half;
          dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Os, B_KLD, q_lo, q_lo + 8, 0), pb, dv0, 0, 0, 0);
          dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Os, B_KLD, q_lo, q_lo + 8, 32), pb, dv1, 0, 0, 0);
          dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Qs, B_KLD, q_lo, q_lo + 8, 0), sb, dk0, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Qs, B_KLD, q_lo, q_lo + 8, 32), sb, dk1, 0, 0, 0);
        }
        // dQ^T[d][i] += K^T[d][j] dS^T[j][i]: keys 16 st + 8 half + 0..7 of the tile; both operands read transposed
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const int k_lo = 16 * st + 8 * half;
          const bf16x8 dsb = tr8(Tw, B_TLD, k_lo, k_lo + 4, 0);          // B: lane (query = col, 8 keys)
          dqa[it][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Kr, B_KLD, k_lo, k_lo + 4, 0), dsb, dqa[it][0], 0, 0, 0);
          dqa[it][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(Kr, B_KLD, k_lo, k_lo + 4, 32), dsb, dqa[it][1], 0, 0, 0);
        }
      }
    }
    // dK / dV of the tile: a lane owns ONE key (accumulator column) and, per register quad, 4 consecutive d: 16-byte pieces
    // (accumulate_dkv: added to what is there -- the keys / values of weight-tied layers collect their gradient over the
    // layers in place, instead of through one 1-GB elementwise add per layer)
    if constexpr (sizeof(DKV) == 2) {
      if (j_ok) {
        uint16_t* dkp = reinterpret_cast<uint16_t*>(dk) + b * g.k_bs + (long long)j * g.k_rs + h * BD;
        uint16_t* dvp = reinterpret_cast<uint16_t*>(dv) + b * g.k_bs + (long long)j * g.k_rs + h * BD;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int d = 8 * q4 + 4 * half;
          *reinterpret_cast<u32x2*>(dkp + d) = (u32x2){pack_bf16_pair(dk0[4 * q4], dk0[4 * q4 + 1]), pack_bf16_pair(dk0[4 * q4 + 2], dk0[4 * q4 + 3])};
          *reinterpret_cast<u32x2*>(dkp + 32 + d) = (u32x2){pack_bf16_pair(dk1[4 * q4], dk1[4 * q4 + 1]), pack_bf16_pair(dk1[4 * q4 + 2], dk1[4 * q4 + 3])};
          *reinterpret_cast<u32x2*>(dvp + d) = (u32x2){pack_bf16_pair(dv0[4 * q4], dv0[4 * q4 + 1]), pack_bf16_pair(dv0[4 * q4 + 2], dv0[4 * q4 + 3])};
          *reinterpret_cast<u32x2*>(dvp + 32 + d) = (u32x2){pack_bf16_pair(dv1[4 * q4], dv1[4 * q4 + 1]), pack_bf16_pair(dv1[4 * q4 + 2], dv1[4 * q4 + 3])};
        }
      }
    } else if (j_ok) {
      float* dkp = reinterpret_cast<float*>(dk) + b * g.k_bs + (long long)j * g.k_rs + h * BD;
      float* dvp = reinterpret_cast<float*>(dv) + b * g.k_bs + (long long)j * g.k_rs + h * BD;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = 8 * q4 + 4 * half;            // att_acc_row(4 q4 + 0..3, half) = d + 0..3
        float4 o[4] = {make_float4(dk0[4 * q4], dk0[4 * q4 + 1], dk0[4 * q4 + 2], dk0[4 * q4 + 3]),
                       make_float4(dk1[4 * q4], dk1[4 * q4 + 1], dk1[4 * q4 + 2], dk1[4 * q4 + 3]),
                       make_float4(dv0[4 * q4], dv0[4 * q4 + 1], dv0[4 * q4 + 2], dv0[4 * q4 + 3]),
                       make_float4(dv1[4 * q4], dv1[4 * q4 + 1], dv1[4 * q4 + 2], dv1[4 * q4 + 3])};
        float4* gp[4] = {reinterpret_cast<float4*>(dkp + d), reinterpret_cast<float4*>(dkp + 32 + d),
                         reinterpret_cast<float4*>(dvp + d), reinterpret_cast<float4*>(dvp + 32 + d)};
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          if (accumulate_dkv) {
            const float4 old = *gp[w4];
            o[w4].x += old.x, o[w4].y += old.y, o[w4].z += old.z, o[w4].w += old.w;
          }
          *gp[w4] = o[w4];
        }
      }
    }
  }
  // ---- add the four waves' dQ^T partials (wave order) and write dq; two query tiles per round through the Q / dO images ----
  // 2 x [wave][query 32][d 64 (+1)] floats: a lane writes its query's column (pitch 65 words: 32 lanes on 32 banks), the sum
  // reads run along d -- the [d][query] form this replaces was read 32-way bank-conflicted (PMC: 84 % of the kernel's LDS
  // cycles were conflict cycles)
  float* scratch = reinterpret_cast<float*>(smem);
  static_assert(2 * 4 * 32 * SLD * 4 <= SMEM_B, "dQ scratch does not fit");
  for (int round = 0; round < 2; ++round) {
    __syncthreads();   // everyone is done with the Q / dO images (first round) or with the previous round's sums
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = 2 * round + u;
      float* dst = scratch + u * (4 * 32 * SLD);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(wave * 32 + col) * SLD + 32 * t + att_acc_row(r, half)] = dqa[it][t][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * 32 * 64; idx += 256) {
      const int u = idx / (32 * 64), rem = idx % (32 * 64);
      const int il = rem / 64, d = rem % 64;
      const int i = (2 * round + u) * 32 + il;
      if (i < g.n_q) {
        const float* src = scratch + u * (4 * 32 * SLD);
        const float sum = ((src[(0 * 32 + il) * SLD + d] + src[(1 * 32 + il) * SLD + d]) + src[(2 * 32 + il) * SLD + d]) +
                          src[(3 * 32 + il) * SLD + d];
        dq[b * g.q_bs + (long long)i * g.q_rs + h * BD + d] = sum;
      }
    }
  }
}

void launch_sum_slabs(const float* slabs, float* out, long long n, int n_slabs, long long stride, long long offset,
                      hipStream_t st, int accumulate = 0);   // gemm_f32.hip
void launch_attn_delta(const float* o, const float* dout, float* delta, const pv_attention_desc* d, hipStream_t st);   // attention_f32.hip

}  // namespace pv

using namespace pv;

extern "C" {

static int attn_geom_b(const pv_attention_desc* d, AttnGeomB* g, const char* who) {
  PV_REQUIRE(d, PV_EINVAL, "%s: null descriptor", who);
  PV_REQUIRE(d->batch > 0 && d->heads > 0 && d->n_q > 0 && d->n_k > 0, PV_EINVAL, "%s: non-positive extent", who);
  PV_REQUIRE(d->head_dim == BD, PV_ESIZE, "%s: head_dim must be %d", who, BD);
  PV_REQUIRE(d->q_row_stride % 4 == 0 && d->k_row_stride % 4 == 0 && d->q_batch_stride % 4 == 0 && d->k_batch_stride % 4 == 0,
             PV_EINVAL, "%s: strides must be multiples of 4 elements (16-byte rows)", who);
  PV_REQUIRE(d->batch <= 65535 && d->heads <= 65535, PV_ESIZE, "%s: batch / heads exceed the grid limit", who);
  g->n_q = d->n_q, g->n_k = d->n_k, g->heads = d->heads;
  g->q_bs = d->q_batch_stride, g->q_rs = d->q_row_stride, g->k_bs = d->k_batch_stride, g->k_rs = d->k_row_stride;
  g->scale = d->scale;
  return PV_OK;
}

static int attn_fwd_splits(const pv_attention_desc* d, int* keys_per_split) {
  const long long groups = (long long)d->batch * d->heads * ((d->n_q + 127) / 128);
  const int n_tiles = (d->n_k + BTJ - 1) / BTJ;
  int splits = (int)((768 + groups - 1) / groups);       // aim at three workgroups per CU (more: slower, measured in round 4)
  if (splits > n_tiles / 16) splits = n_tiles / 16;       // at least 16 key tiles (512 keys) per workgroup
  if (splits < 1) splits = 1;
  const int per = (n_tiles + splits - 1) / splits;
  *keys_per_split = per * BTJ;
  return (n_tiles + per - 1) / per;
}

size_t pv_attention_fwd_workspace_floats(const pv_attention_desc* d) {
  if (!d || d->batch <= 0 || d->heads <= 0 || d->n_q <= 0 || d->n_k <= 0) return 0;
  int per;
  const int nsp = attn_fwd_splits(d, &per);
  return nsp > 1 ? (size_t)nsp * d->batch * d->heads * d->n_q * (BD + 2) : 0;
}

extern "C++" template <typename KV>
static int attention_fwd_bf16_any(const float* q, const KV* k, const KV* v, float* o, float* lse, const pv_attention_desc* d,
                                  float* workspace, void* stream) {
  AttnGeomB g;
  int rc = attn_geom_b(d, &g, "pv_attention_fwd_bf16");
  if (rc) return rc;
  PV_REQUIRE(sizeof(KV) == 4 || (d->k_row_stride % 8 == 0 && d->k_batch_stride % 8 == 0), PV_EINVAL,
             "pv_attention_fwd_bf16kv: K / V strides must be multiples of 8 elements (16-byte rows)");
  PV_REQUIRE(q && k && v && o && lse, PV_EINVAL, "pv_attention_fwd_bf16: null pointer");
  PV_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0), PV_EINVAL,
             "pv_attention_fwd_bf16: q / k / v must be 16-byte aligned");
  int per = d->n_k;
  const int nsp = workspace ? attn_fwd_splits(d, &per) : 1;     // no workspace: one workgroup walks all keys
  const int n_qt = (d->n_q + 127) / 128;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)(n_qt * nsp), (unsigned)d->heads, (unsigned)d->batch);
  hipLaunchKernelGGL(attn_fwd_bf16<KV>, grid, dim3(256), 0, st, q, k, v, o, lse, g, nsp, nsp > 1 ? per : d->n_k, workspace);
  if (nsp > 1) {
    const long long rows = (long long)d->batch * d->heads * d->n_q;
    hipLaunchKernelGGL(attn_fwd_combine, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const float*)workspace, o, lse, g,
                       nsp, rows, d->batch);
  }
  return check_launch("pv_attention_fwd_bf16");
}

int pv_attention_fwd_bf16(const float* q, const float* k, const float* v, float* o, float* lse, const pv_attention_desc* d,
                          float* workspace, void* stream) {
  return attention_fwd_bf16_any<float>(q, k, v, o, lse, d, workspace, stream);
}

int pv_attention_fwd_bf16kv(const float* q, const uint16_t* k, const uint16_t* v, float* o, float* lse, const pv_attention_desc* d,
                            float* workspace, void* stream) {
  return attention_fwd_bf16_any<uint16_t>(q, k, v, o, lse, d, workspace, stream);
}

extern "C++" template <typename KV, typename DKV>
static int attention_bwd_bf16_any(const float* q, const KV* k, const KV* v, const float* o, const float* dout, const float* lse,
                                  float* delta_ws, float* dq, DKV* dk, DKV* dv, const pv_attention_desc* d,
                                  int32_t accumulate_dkv, void* stream) {
  PV_REQUIRE(sizeof(DKV) == 4 || !accumulate_dkv, PV_EINVAL, "pv_attention_bwd_bf16kv16: bf16 dK / dV cannot be accumulated into");
  AttnGeomB g;
  int rc = attn_geom_b(d, &g, "pv_attention_bwd_bf16");
  if (rc) return rc;
  PV_REQUIRE(sizeof(KV) == 4 || (d->k_row_stride % 8 == 0 && d->k_batch_stride % 8 == 0), PV_EINVAL,
             "pv_attention_bwd_bf16kv: K / V strides must be multiples of 8 elements (16-byte rows)");
  PV_REQUIRE(q && k && v && o && dout && lse && delta_ws && dq && dk && dv, PV_EINVAL, "pv_attention_bwd_bf16: null pointer");
  PV_REQUIRE(d->n_q <= 128, PV_ESIZE, "pv_attention_bwd_bf16: n_q=%d > 128 queries per (batch, head) is not built", d->n_q);
  PV_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                 ((uintptr_t)dk % 16 == 0) && ((uintptr_t)dv % 16 == 0),
             PV_EINVAL, "pv_attention_bwd_bf16: q / dout / k / v / dk / dv must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const long long rows = (long long)d->batch * d->heads * d->n_q;
  launch_attn_delta(o, dout, delta_ws, d, st);    // the row term sum_d dO * O stays f32 (same kernel as the f32 path)
  // key splits exactly as pv_attention_bwd_f32 / pv_attention_bwd_workspace_floats lay them out
  const long long groups = (long long)d->batch * d->heads;
  const int n_tiles = (d->n_k + BTJ - 1) / BTJ;
  // whole rounds of one workgroup per CU (102 KB of LDS each): at most 3 x 256 workgroups -- every workgroup pays the
  // staging of Q / dO in two layouts and the dQ reduction once, so fewer, longer key ranges win over a ragged fifth round.
  // Never more splits than pv_attention_bwd_workspace_floats (attention_f32.hip) provides for.
  int splits = (int)(768 / groups);
  if (splits > n_tiles / 8) splits = n_tiles / 8;
  if (splits < 1) splits = 1;
  const int per = ((n_tiles + splits - 1) / splits + 3) / 4 * 4;
  const int nsp = (n_tiles + per - 1) / per;
  if (nsp > 1) {
    float* part = delta_ws + rows;
    const long long n = (long long)d->batch * d->q_batch_stride;
    hipLaunchKernelGGL((attn_bwd_bf16<KV, DKV>), dim3((unsigned)nsp, (unsigned)d->heads, (unsigned)d->batch), dim3(256), 0, st, q, k, v, dout,
                       lse, (const float*)delta_ws, part, dk, dv, g, per, n, accumulate_dkv);
    launch_sum_slabs(part, dq, n, nsp, n, 0, st);
  } else {
    hipLaunchKernelGGL((attn_bwd_bf16<KV, DKV>), dim3(1, (unsigned)d->heads, (unsigned)d->batch), dim3(256), 0, st, q, k, v, dout, lse,
                       (const float*)delta_ws, dq, dk, dv, g, n_tiles, 0ll, accumulate_dkv);
  }
  return check_launch("pv_attention_bwd_bf16");
}

int pv_attention_bwd_bf16(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                          float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, int32_t accumulate_dkv,
                          void* stream) {
  return attention_bwd_bf16_any<float, float>(q, k, v, o, dout, lse, delta_ws, dq, dk, dv, d, accumulate_dkv, stream);
}

int pv_attention_bwd_bf16kv(const float* q, const uint16_t* k, const uint16_t* v, const float* o, const float* dout, const float* lse,
                            float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, int32_t accumulate_dkv,
                            void* stream) {
  return attention_bwd_bf16_any<uint16_t, float>(q, k, v, o, dout, lse, delta_ws, dq, dk, dv, d, accumulate_dkv, stream);
}

int pv_attention_bwd_bf16kv16(const float* q, const uint16_t* k, const uint16_t* v, const float* o, const float* dout,
                              const float* lse, float* delta_ws, float* dq, uint16_t* dk, uint16_t* dv, const pv_attention_desc* d,
                              void* stream) {
  return attention_bwd_bf16_any<uint16_t, uint16_t>(q, k, v, o, dout, lse, delta_ws, dq, dk, dv, d, 0, stream);
}

#ifdef PV_DIAG_STAMPS
int pv_diag_read_attn_fwd(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::attn_fwd_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
