// LDS-tiled fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]
//
// Exact fp32 (the MFMA is bitwise an fmaf chain), which is what the parity
// bar needs: greedy NA-decode tokens must match the reference's fp32 CPU path.
//
// Operand layouts (per operand, compile time):
//   KC ("reduce-dim contiguous"):  element (row, k) at base[row*ld + k]
//   MC ("row-dim contiguous"):     element (row, k) at base[k*ld + row]
// so  y = x W^T      is (Q=x  KC, P=W  KC)       [forward, nn.Linear]
//     dx = dz W      is (Q=dz KC, P=W  MC)       [reduce over n]
//     dW = dz^T x    is (Q=dz MC, P=x  MC)       [reduce over m; Q rows = n]
//
// Row sets ("varlen" without repacking): the activation rows of the decoder are
// [sequence, position] slots of which 40-70 % are <pad>.  A GEMM can be handed a
// device-side list of the live rows (`rows`, `count`): the activation-row index
// of the tile walk is then an index INTO that list, operands are gathered and
// results scattered through it, and tiles beyond `count` exit at once.  Buffers
// keep their dense [slots, D] layout, so nothing else in the pipeline changes,
// shapes stay static (hipGraph-safe) and the host never learns the count.
//   NT / NN: Q rows and C rows go through the list;   TN: the reduce index does.
//
// Workgroup: 256 threads = 4 waves (WM x WN), block tile BM x BN, BK = 16,
// two LDS images + prefetched fragment registers + global->register staging
// three k-tiles ahead (one barrier per k-tile, see "software pipeline" in the
// kernel).  Wave tile = (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 16x16.
//
// LDS images and the k-permutation trick: the 16x16x4 MFMA wants, from lane
// (i = lane&15, g = lane>>4), A[i][k=g].  Which physical k each (step, g)
// pair uses is free as long as both operands agree, so step s of a 16-deep
// k-tile uses k = 4g + s:
//   KC tiles are stored [row][16] (64 B rows, as they arrive from HBM) and a
//      lane fetches its 4 steps with ONE ds_read_b128 of columns 4g..4g+3; the
//      16-B column slot is XOR-swizzled with sw(row) = (-(row>>2))&3, which
//      makes every ds_read_b128 lane group {rows x g} hit 16 distinct slots
//      and keeps the staging ds_write_b128 conflict-free as well
//      (SQ_LDS_BANK_CONFLICT = 0 measured);
//   MC tiles are stored [16][rows+4] (row stride = 4 mod 8 dwords) and read
//      with ds_read_b32 at [(4g+s)][row]: the two g's of a 32-lane group land
//      16 banks apart.
// The MFMA is issued "swapped" (a = P fragment, b = Q fragment) so each lane
// ends up with 4 CONSECUTIVE n of one row m: epilogues store float4s.
#pragma once
#include "common.hpp"
#include <type_traits>

// Tuning aid, compiled in only with -DNACF_GEMM_TRACE (tools/gemm_trace.py): wave 0 of every workgroup records the
// shader clock at kernel entry / after the prologue / after the k-loop / at the end, plus where it ran.
#ifdef NACF_GEMM_TRACE
__device__ unsigned long long* g_gemm_trace = nullptr;
#define NACF_TRACE_MARK(slot)                                                                               \
  do {                                                                                                      \
    if (g_gemm_trace && threadIdx.x == 0)                                                                   \
      g_gemm_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define NACF_TRACE_MARK(slot) do { } while (0)
#endif

struct GemmShape {
  const float* Q;
  const float* P;
  int64_t ldq, ldp;
  int M, N, K;
  int k_per_split;  // multiple of 16
  int tiles_m, tiles_n;
  const int* rows;   // optional live-row list (see "Row sets" above)
  const int* count;  // device int: number of valid entries of `rows`
  int zero_dead;     // NT / NN with a row list: rows[count..M) are the dead rows; their output rows are zero-filled
  // TN only (dW = dZ^T X): the bias gradient db[n] = sum_m dZ[m][n] is the row sum of the Q operand this GEMM
  // stages anyway.  The workgroups of column tile 0 add up what they stage: straight into `colsum_out`
  // (= beta * old + sum) without a reduce split, else into colsum_part[z][n] for the split-K combine kernel.
  float* colsum_part;
  float* colsum_out;
  float colsum_beta;
  // bf16 matrix-core kernels only (gemm_bf16.hpp): the P operand as pre-converted bf16 image(s), K-contiguous rows
  // (element (row, k) of split plane s at Pimg[s * pimg_plane + row * ldpi + k]); nullptr = convert P in the kernel
  const unsigned short* Pimg;
  int64_t ldpi;
  int64_t pimg_plane;
  // bf16 kernels: tile order inside an XCD's run.  0: n fastest over ALL column tiles (one row tile's X stays in L2, the
  // whole of P streams past once per row tile).  g > 0: column tiles in groups of g, inside a group n fastest, then m --
  // a group's slice of P (g * BN rows) stays in the XCD's 4 MB L2 while X streams past once per group: for a wide P
  // (the vocabulary: 83 column tiles) that is tiles_n / g passes over X instead of tiles_m / 8 passes over P.
  int group_n = 0;
};

__device__ __forceinline__ int lds_sw(int row) { return (-(row >> 2)) & 3; }

// KC tile: `prow[u]` = physical row of this thread's u-th vector (or -1: out of range -> zeros)
template <int R, bool VEC>
__device__ __forceinline__ void tile_load_kc(f32x4 (&reg)[R / 64], const float* __restrict__ base, int64_t ld,
                                             const int (&prow)[R / 64], int k0, int k_end, int tid) {
#pragma unroll
  for (int u = 0; u < R / 64; ++u) {
    const int c = (tid + 256 * u) & 3;
    const int gk = k0 + c * 4;
    const int nv = k_end - gk;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (prow[u] >= 0) {
      const float* p = base + (int64_t)prow[u] * ld + gk;
      if (nv >= 4) {
        if (VEC) {
          t = *reinterpret_cast<const f32x4*>(p);
        } else {
          t[0] = p[0]; t[1] = p[1]; t[2] = p[2]; t[3] = p[3];
        }
      } else {
        if (nv > 0) t[0] = p[0];
        if (nv > 1) t[1] = p[1];
        if (nv > 2) t[2] = p[2];
      }
    }
    reg[u] = t;
  }
}

// MC tile: rows are the contiguous dimension; the reduce index k may go through `kmap`
template <int R, bool VEC>
__device__ __forceinline__ void tile_load_mc(f32x4 (&reg)[R / 64], const float* __restrict__ base, int64_t ld,
                                             int row0, int rows_total, int k0, int k_end,
                                             const int* __restrict__ kmap, int tid) {
  constexpr int RV = R / 4;
#pragma unroll
  for (int u = 0; u < R / 64; ++u) {
    const int q = tid + 256 * u;
    const int kk = q / RV, r4 = q % RV;
    const int gk = k0 + kk, gr = row0 + r4 * 4;
    const int nv = rows_total - gr;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (gk < k_end) {
      const int pk = kmap ? kmap[gk] : gk;
      const float* p = base + (int64_t)pk * ld + gr;
      if (nv >= 4) {
        if (VEC) {
          t = *reinterpret_cast<const f32x4*>(p);
        } else {
          t[0] = p[0]; t[1] = p[1]; t[2] = p[2]; t[3] = p[3];
        }
      } else {
        if (nv > 0) t[0] = p[0];
        if (nv > 1) t[1] = p[1];
        if (nv > 2) t[2] = p[2];
      }
    }
    reg[u] = t;
  }
}

template <int R, bool KC>
__device__ __forceinline__ void tile_store(float* lds, const f32x4 (&reg)[R / 64], int tid) {
#pragma unroll
  for (int u = 0; u < R / 64; ++u) {
    const int q = tid + 256 * u;
    if (KC) {
      const int rr = q >> 2, c = q & 3;
      *reinterpret_cast<f32x4*>(&lds[rr * 16 + ((c ^ lds_sw(rr)) << 2)]) = reg[u];
    } else {
      constexpr int RV = R / 4;
      const int kk = q / RV, r4 = q % RV;
      *reinterpret_cast<f32x4*>(&lds[kk * (R + 4) + r4 * 4]) = reg[u];
    }
  }
}

// KC fragment: the 4 k-steps of one 16-row MFMA tile in one ds_read_b128
template <int R>
__device__ __forceinline__ f32x4 frag_load_kc(const float* lds, int rb, int i, int g) {
  const int row = rb + i;
  return *reinterpret_cast<const f32x4*>(&lds[row * 16 + ((g ^ lds_sw(row)) << 2)]);
}
// MC fragment: ONE k-step of all T 16-row MFMA tiles of the wave in one ds_read_b128 (T = 4) / b64 (T = 2).
// The MC image is [k][rows], so what is contiguous for a lane is a run of ROWS; the wave tile's rows are
// therefore dealt to the MFMA tiles round-robin -- MFMA tile t, row index i  <->  wave-tile row T*i + t --
// and the lane reads rows T*i .. T*i+T-1 of image row k = 4g+s.  (A per-tile layout would need one
// ds_read_b32 per (tile, k-step): 4x the LDS instructions.)  The accumulator -> (m, n) map changes with it,
// see "accumulator map" in the kernel.
template <int N>
using fvec = float __attribute__((ext_vector_type(N)));
template <int R, int T>
__device__ __forceinline__ fvec<T> frag_load_mc(const float* lds, int rb, int i, int g, int s) {
  return *reinterpret_cast<const fvec<T>*>(&lds[(g * 4 + s) * (R + 4) + rb + T * i]);
}

// ---------------------------------------------------------------- epilogues
// float4 of 4 consecutive output columns, slot j of row group a (see "accumulator map" in the kernel): with a KC
// P operand that is acc[a][j]; with an MC P the 4 columns live in the TN different MFMA tiles.
template <bool PKC, int TM, int TN>
__device__ __forceinline__ f32x4 out_vec(const f32x4 (&acc)[TM][TN], int a, int j) {
  if constexpr (PKC) return acc[a][j];
  else if constexpr (TN == 4) return f32x4{acc[a][0][j], acc[a][1][j], acc[a][2][j], acc[a][3][j]};
  else return f32x4{acc[a][0][2 * j], acc[a][1][2 * j], acc[a][0][2 * j + 1], acc[a][1][2 * j + 1]};
}

// Epilogue call: (m = logical row for guards, mp = physical row for addressing, n, 4 values).
// plain / accumulate / split-K slab store
struct EpiStore {
  float* C;
  int64_t ldc;
  float beta;
  int64_t slab_stride;  // elements between blockIdx.z slabs (0: none)
  int vec_out;
  static constexpr bool kArgmax = false;
  __device__ __forceinline__ void operator()(int m, int mp, int n, f32x4 v, int M, int N, int z) const {
    if (m >= M || n >= N) return;
    float* p = C + (int64_t)z * slab_stride + (int64_t)mp * ldc + n;
    const int nv = N - n;
    if (nv >= 4 && vec_out) {
      if (beta != 0.f) {
        f32x4 o = *reinterpret_cast<const f32x4*>(p);
        v[0] += beta * o[0]; v[1] += beta * o[1]; v[2] += beta * o[2]; v[3] += beta * o[3];
      }
      *reinterpret_cast<f32x4*>(p) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nv) p[e] = (beta != 0.f) ? v[e] + beta * p[e] : v[e];
    }
  }
  // dead-row fill (see nacf_rowset.zero_dead): columns [n, n + 4) of physical row mp
  __device__ __forceinline__ void zero4(int mp, int n, int N) const {
    if (beta != 0.f || slab_stride != 0) return;   // accumulate: dead rows contribute nothing
    float* p = C + (int64_t)mp * ldc + n;
    if (vec_out && n + 4 <= N) *reinterpret_cast<f32x4*>(p) = f32x4{0.f, 0.f, 0.f, 0.f};
    else
      for (int e = 0; e < 4; ++e)
        if (n + e < N) p[e] = 0.f;
  }
  // interior tile (every row < M, every column group < N, float4-addressable): all loads are issued
  // before the first store, nothing is guarded.  The per-call path above serialises one load -> wait ->
  // store chain per 16x16 sub-tile, which made the epilogue of a 128x128 tile cost ~8 % of a K=512 GEMM.
  __device__ __forceinline__ bool fast_ok() const { return vec_out != 0; }
  template <int TM, int TN, bool PKC>
  __device__ __forceinline__ void tile_fast(f32x4 (&acc)[TM][TN], const int (&mphys)[TM], const int (&ncol)[TN], int, int z) const {
    float* base = C + (int64_t)z * slab_stride;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      float* row = base + (int64_t)mphys[a] * ldc;
      if (beta != 0.f) {
        f32x4 old[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) old[b] = *reinterpret_cast<const f32x4*>(row + ncol[b]);
#pragma unroll
        for (int b = 0; b < TN; ++b) *reinterpret_cast<f32x4*>(row + ncol[b]) = out_vec<PKC>(acc, a, b) + beta * old[b];
      } else {
#pragma unroll
        for (int b = 0; b < TN; ++b) *reinterpret_cast<f32x4*>(row + ncol[b]) = out_vec<PKC>(acc, a, b);
      }
    }
  }
};

// nn.Linear forward epilogue (see nacf_epilogue in nacf_hip.h)
struct EpiLinear {
  float* Y;
  int64_t ldy;
  nacf_epilogue ep;
  int vec_out;   // Y / preact / residual all float4-addressable
  int vec_bias;  // bias float4-addressable (or absent); N % 4 == 0 when a dropout is on (aligned Philox groups)
  static constexpr bool kArgmax = false;
  __device__ __forceinline__ void operator()(int m, int mp, int n, f32x4 v, int M, int N, int /*z*/) const {
    if (m >= M || n >= N) return;
    const int nv = N - n;
    if (ep.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nv) v[e] += ep.bias[n + e];
    }
    if (ep.preact) {
      float* z = ep.preact + (int64_t)mp * ep.ld_preact + n;
      if (nv >= 4 && vec_out) *reinterpret_cast<f32x4*>(z) = v;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nv) z[e] = v[e];
      }
    }
    if (ep.act != NACF_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(ep.act, v[e], n + e, ep.act_split);
    }
    const bool any_drop = (ep.p_drop1 > 0.f) || (ep.p_drop2 > 0.f);
    DropRng rng;
    if (any_drop) rng.init(ep.rng_state);
    const uint64_t e0 = (uint64_t)mp * (uint64_t)N + (uint64_t)n;   // physical slot: masks do not depend on packing
    const bool grp_ok = (N & 3) == 0;  // aligned groups of 4 share one Philox call
    if (ep.p_drop1 > 0.f) {
      if (grp_ok) {
        f32x4 k = rng.keep4(e0 >> 2, ep.salt1, ep.p_drop1);
        v[0] *= k[0]; v[1] *= k[1]; v[2] *= k[2]; v[3] *= k[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= rng.keep1(e0 + e, ep.salt1, ep.p_drop1);
      }
    }
    if (ep.residual) {
      const float* r = ep.residual + (int64_t)mp * ep.ld_residual + n;
      if (nv >= 4 && vec_out) {
        f32x4 o = *reinterpret_cast<const f32x4*>(r);
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nv) v[e] += r[e];
      }
    }
    if (ep.p_drop2 > 0.f) {
      if (grp_ok) {
        f32x4 k = rng.keep4(e0 >> 2, ep.salt2, ep.p_drop2);
        v[0] *= k[0]; v[1] *= k[1]; v[2] *= k[2]; v[3] *= k[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= rng.keep1(e0 + e, ep.salt2, ep.p_drop2);
      }
    }
    if (ep.row_tokens) {
      if (ep.row_tokens[mp] == NACF_PAD) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
    }
    float* y = Y + (int64_t)mp * ldy + n;
    if (nv >= 4 && vec_out) *reinterpret_cast<f32x4*>(y) = v;
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nv) y[e] = v[e];
    }
  }
  // dead-row fill (see nacf_rowset.zero_dead): Y and the saved pre-activation
  __device__ __forceinline__ void zero4(int mp, int n, int N) const {
    float* y = Y + (int64_t)mp * ldy + n;
    float* z = ep.preact ? ep.preact + (int64_t)mp * ep.ld_preact + n : nullptr;
    if (vec_out && n + 4 <= N) {
      *reinterpret_cast<f32x4*>(y) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (z) *reinterpret_cast<f32x4*>(z) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      for (int e = 0; e < 4; ++e)
        if (n + e < N) { y[e] = 0.f; if (z) z[e] = 0.f; }
    }
  }
  // interior tile: bias / residual / row-token loads are all issued up front (see EpiStore::tile_fast)
  __device__ __forceinline__ bool fast_ok() const { return vec_out != 0 && vec_bias != 0; }
  template <int TM, int TN, bool PKC>
  __device__ __forceinline__ void tile_fast(f32x4 (&acc)[TM][TN], const int (&mphys)[TM], const int (&ncol)[TN], int N, int) const {
    static_assert(PKC, "nn.Linear forward is an NT GEMM");
    f32x4 bias[TN];
    f32x4 res[TM][TN];
    bool dead[TM];
#pragma unroll
    for (int b = 0; b < TN; ++b)
      bias[b] = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + ncol[b]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      dead[a] = ep.row_tokens ? (ep.row_tokens[mphys[a]] == NACF_PAD) : false;
#pragma unroll
      for (int b = 0; b < TN; ++b)
        res[a][b] = ep.residual ? *reinterpret_cast<const f32x4*>(ep.residual + (int64_t)mphys[a] * ep.ld_residual + ncol[b])
                                : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool any_drop = (ep.p_drop1 > 0.f) || (ep.p_drop2 > 0.f);
    DropRng rng;
    if (any_drop) rng.init(ep.rng_state);
    // ONE uniform dispatch on (activation, dropout) up front, then a straight-line body per combination: with the
    // tests inside the TM x TN unrolled body every sub-tile drags its own copy of every activation and of the Philox
    // rounds along (72 k instructions for a 128 x 128 tile), and the walk over those mostly skipped copies costs the
    // workgroup ~14 k cycles of instruction-cache misses -- as much as four k-tiles of matrix work.
    if (!any_drop) {
      switch (ep.act) {
        case NACF_ACT_NONE: fast_body<0, TM, TN, NACF_ACT_NONE, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        case NACF_ACT_RELU: fast_body<0, TM, TN, NACF_ACT_RELU, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        case NACF_ACT_GELU_NEW: fast_body<0, TM, TN, NACF_ACT_GELU_NEW, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        case NACF_ACT_TANH: fast_body<0, TM, TN, NACF_ACT_TANH, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        case NACF_ACT_SIGMOID: fast_body<0, TM, TN, NACF_ACT_SIGMOID, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        case NACF_ACT_TANH_SIGMOID: fast_body<0, TM, TN, NACF_ACT_TANH_SIGMOID, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
        default: fast_body<0, TM, TN, NACF_ACT_GELU_ERF, false>(acc, mphys, ncol, N, bias, res, dead, rng); break;
      }
    } else if (ep.act == NACF_ACT_NONE) {
      fast_body<0, TM, TN, NACF_ACT_NONE, true>(acc, mphys, ncol, N, bias, res, dead, rng);
    } else if (ep.act == NACF_ACT_RELU) {
      fast_body<0, TM, TN, NACF_ACT_RELU, true>(acc, mphys, ncol, N, bias, res, dead, rng);
    } else {
      fast_body<0, TM, TN, -1, true>(acc, mphys, ncol, N, bias, res, dead, rng);      // rare: every test stays inside
    }
  }
  // compile-time recursion over the TM x TN sub-tiles (a `#pragma unroll` loop over this body is declined by
  // the optimiser and the register arrays end up in scratch).  ACT: the activation, or -1 = read ep.act per element.
  template <int IDX, int TM, int TN, int ACT, bool DROP>
  __device__ __forceinline__ void fast_body(f32x4 (&acc)[TM][TN], const int (&mphys)[TM], const int (&ncol)[TN], int N,
                                            const f32x4 (&bias)[TN], const f32x4 (&res)[TM][TN], const bool (&dead)[TM],
                                            const DropRng& rng) const {
    if constexpr (IDX < TM * TN) {
      constexpr int a = IDX / TN, b = IDX % TN;
      const int n = ncol[b];
      f32x4 v = acc[a][b] + bias[b];
      if (ep.preact) *reinterpret_cast<f32x4*>(ep.preact + (int64_t)mphys[a] * ep.ld_preact + n) = v;
      if constexpr (ACT != NACF_ACT_NONE) {
        const int act = ACT < 0 ? ep.act : ACT;
        if (ACT >= 0 || act != NACF_ACT_NONE) {
          v[0] = apply_act(act, v[0], n, ep.act_split);
          v[1] = apply_act(act, v[1], n + 1, ep.act_split);
          v[2] = apply_act(act, v[2], n + 2, ep.act_split);
          v[3] = apply_act(act, v[3], n + 3, ep.act_split);
        }
      }
      if constexpr (DROP) {
        const uint64_t grp = ((uint64_t)mphys[a] * (uint64_t)N + (uint64_t)n) >> 2;
        if (ep.p_drop1 > 0.f) v *= rng.keep4(grp, ep.salt1, ep.p_drop1);
        v += res[a][b];
        if (ep.p_drop2 > 0.f) v *= rng.keep4(grp, ep.salt2, ep.p_drop2);
      } else {
        v += res[a][b];
      }
      if (dead[a]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Y + (int64_t)mphys[a] * ldy + n) = v;
      fast_body<IDX + 1, TM, TN, ACT, DROP>(acc, mphys, ncol, N, bias, res, dead, rng);
    }
  }
};

// fused "softmax-max" partials: per (n-tile, logical row): max logit, its index, sum exp(l - max)
struct EpiArgmax {
  const float* bias;
  float* pmax;   // [tiles_n][M]
  float* psum;   // [tiles_n][M]
  int* pidx;     // [tiles_n][M]
  // training form (nacf_vocab_lse_fwd): the logits are ALSO stored (C[phys row][n], ldc % 4 == 0, 16-byte aligned) and
  // the sum-exp uses expf; decode (C == nullptr) never materialises them and uses the fast exp
  float* C = nullptr;
  int64_t ldc = 0;
  static constexpr bool kArgmax = true;
  __device__ __forceinline__ void operator()(int, int, int, f32x4, int, int, int) const {}
  __device__ __forceinline__ void zero4(int, int, int) const {}
};

template <int IDX, int TM, int TN, bool PKC, class Epi>
__device__ __forceinline__ void epilogue_all(const Epi& epi, f32x4 (&acc)[TM][TN], const int (&mlog)[TM], const int (&mphys)[TM],
                                             const int (&ncol)[TN], int M, int N, int z) {
  if constexpr (IDX < TM * TN) {
    constexpr int a = IDX / TN, b = IDX % TN;
    epi(mlog[a], mphys[a], ncol[b], out_vec<PKC>(acc, a, b), M, N, z);
    epilogue_all<IDX + 1, TM, TN, PKC, Epi>(epi, acc, mlog, mphys, ncol, M, N, z);
  }
}

// ---------------------------------------------------------------- fused soft-max statistics (EpiArgmax)
// per-row (max, argmax, sum-exp) over this workgroup tile's BN columns; accumulator map: acc[a][b][e] is row
// wm*WTM + a*16 + li, column wn*WTN + b*16 + lg*4 + e of the tile (KC / KC operands).  `smem` is scratch of at least
// 3*WN*BM floats that no wave still reads as a tile image (the callers end their main loop with a barrier).
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void argmax_epilogue(float* smem, const GemmShape& g, const EpiArgmax& epi, f32x4 (&acc)[TM][TN],
                                                int m0, int n0, int Meff, int tile_n, int wm, int wn, int li, int lg, int tid) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  {
    // per-row (max, argmax, sum-exp) over this tile's BN columns
    float* redv = smem;                  // [WN][BM]
    float* reds = smem + WN * BM;        // [WN][BM]
    int* redi = reinterpret_cast<int*>(smem + 2 * WN * BM);
    const float NEG = -3.0e38f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      float best = NEG;
      int bidx = 0x7fffffff;
      const int mrow = m0 + wm * WTM + a * 16 + li;                      // logical row of this lane's accumulators
      float* crow = nullptr;
      if (epi.C && mrow < Meff) crow = epi.C + (int64_t)(g.rows ? g.rows[mrow] : mrow) * epi.ldc;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + wn * WTN + b * 16 + lg * 4 + e;
          float v = NEG;
          if (n < g.N) {
            v = acc[a][b][e] + (epi.bias ? epi.bias[n] : 0.f);
            acc[a][b][e] = v;
          } else {
            acc[a][b][e] = NEG;
          }
          if (v > best) { best = v; bidx = n; }
        }
        if (crow) {
          const int nb = n0 + wn * WTN + b * 16 + lg * 4;
          if (nb + 3 < g.N) *reinterpret_cast<f32x4*>(crow + nb) = acc[a][b];
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < g.N) crow[nb + e] = acc[a][b][e];
          }
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
      }
      const int row = wm * WTM + a * 16 + li;
      if (lg == 0) { redv[wn * BM + row] = best; redi[wn * BM + row] = bidx; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int row = wm * WTM + a * 16 + li;
      float tmax = redv[row];
#pragma unroll
      for (int w = 1; w < WN; ++w) tmax = fmaxf(tmax, redv[w * BM + row]);
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[a][b][e];
          s += (v > -1.0e38f) ? (epi.C ? expf(v - tmax) : __expf(v - tmax)) : 0.f;
        }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lg == 0) reds[wn * BM + row] = s;
    }
    __syncthreads();
    for (int row = tid; row < BM; row += 256) {
      const int m = m0 + row;
      if (m >= Meff) continue;
      float best = redv[row];
      int bidx = redi[row];
      float s = reds[row];
#pragma unroll
      for (int w = 1; w < WN; ++w) {
        float ov = redv[w * BM + row];
        int oi = redi[w * BM + row];
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        s += reds[w * BM + row];
      }
      const int64_t o = (int64_t)tile_n * g.M + m;
      epi.pmax[o] = best;
      epi.psum[o] = s;
      epi.pidx[o] = bidx;
    }
  }
}

// ---------------------------------------------------------------- kernel
template <int BM, int BN, int WM, int WN, bool QKC, bool PKC, bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmShape g, Epi epi) {
  constexpr int BK = 16;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int QSZ = QKC ? BM * BK : BK * (BM + 4);
  constexpr int PSZ = PKC ? BN * BK : BK * (BN + 4);
  constexpr int BUF = QSZ + PSZ;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile loader granularity");
  constexpr int RED = Epi::kArgmax ? 3 * WN * BM : 0;
  constexpr int SMEM = (2 * BUF > RED) ? 2 * BUF : RED;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  NACF_TRACE_MARK(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;

  // XCD-aware tile order: workgroup b runs on XCD b%8; give every XCD a
  // contiguous run of logical tiles (n fastest) so the tiles that share a
  // Q row panel hit the same private L2 (bijective for any grid size).
  // live-row list: NT/NN bound the row walk, TN bounds the reduce walk
  constexpr bool ROWS_ARE_K = !QKC && !PKC;
  int Meff = g.M, Keff = g.K;
  if (g.count) {
    const int c = *g.count;
    if (ROWS_ARE_K) Keff = min(Keff, c);
    else Meff = min(Meff, c);
  }
  // only the first (live row tiles x tiles_n) workgroups have work; the remap below is done over
  // THAT count so the live tiles spread over all 8 XCDs (dead row tiles would otherwise idle the
  // XCDs whose contiguous chunk of the tile list lies beyond the live rows)
  const int tiles_m_live = (Meff + BM - 1) / BM;
  const int nwg = tiles_m_live * g.tiles_n;
  const int bid = blockIdx.x;
  if (bid >= nwg) {
    // workgroups past the live tiles: nothing to multiply.  With zero_dead they zero-fill the dead rows
    // (rows[Meff .. M)) of the output instead, BM rows x BN columns each, so no memset is needed.
    if constexpr (!ROWS_ARE_K) {
      if (g.zero_dead && g.rows && blockIdx.z == 0) {
        const int dt = bid - nwg;
        const int j0 = (dt / g.tiles_n) * BM, c0 = (dt % g.tiles_n) * BN;
        const int n_dead = g.M - Meff;
        for (int q = tid; q < BM * (BN / 4); q += 256) {
          const int j = j0 + q / (BN / 4), n = c0 + (q % (BN / 4)) * 4;
          if (j < n_dead && n < g.N) epi.zero4(g.rows[Meff + j], n, g.N);
        }
      }
    }
    return;
  }
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
  const int tile_m = logical / g.tiles_n, tile_n = logical % g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int z = blockIdx.z;
  int kps = g.k_per_split;
  if (ROWS_ARE_K && g.count && gridDim.z > 1) {
    // re-balance the reduce-dimension splits over the LIVE rows only
    kps = (((Keff + (int)gridDim.z - 1) / (int)gridDim.z) + BK - 1) / BK * BK;
    if (kps < BK) kps = BK;
  }
  const int kbeg = z * kps;
  const int kend = min(Keff, kbeg + kps);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  // physical rows of this thread's K-contiguous vectors (fixed for the whole k-loop).  `*row` = -1 marks a row
  // beyond the edge for the checked loader; the fast loader reads the CLAMPED row instead - whatever that
  // produces lands in accumulator rows / columns every epilogue discards (m >= Meff, n >= N).
  int qrow[BM / 64], prow[BN / 64];
  const float* qfast[BM / 64];
  const float* pfast[BN / 64];
#pragma unroll
  for (int u = 0; u < BM / 64; ++u) {
    const int q = tid + 256 * u;
    if constexpr (QKC) {
      const int gr = m0 + (q >> 2);
      const int gc = min(gr, Meff - 1);
      const int ph = g.rows ? g.rows[gc] : gc;
      qrow[u] = (gr < Meff) ? ph : -1;
      qfast[u] = g.Q + (int64_t)ph * g.ldq + (q & 3) * 4;
    } else {
      qrow[u] = 0;
      qfast[u] = g.Q + m0 + (q % (BM / 4)) * 4;
    }
  }
#pragma unroll
  for (int u = 0; u < BN / 64; ++u) {
    const int q = tid + 256 * u;
    if constexpr (PKC) {
      const int gr = n0 + (q >> 2);
      const int gc = min(gr, g.N - 1);
      prow[u] = (gr < g.N) ? gr : -1;
      pfast[u] = g.P + (int64_t)gc * g.ldp + (q & 3) * 4;
    } else {
      prow[u] = 0;
      pfast[u] = g.P + n0 + (q % (BN / 4)) * 4;
    }
  }
  const int* kmap = ROWS_ARE_K ? g.rows : nullptr;
  // row-contiguous (MC) operands take the unchecked loader only when the whole row tile is inside the matrix
  const bool q_full = QKC || (m0 + BM <= Meff);
  const bool p_full = PKC || (n0 + BN <= g.N);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- global -> register staging of one k-tile.  Interior k-tiles (block-uniform test) use plain
  // 16-byte loads with no per-lane guards; the last partial k-tile and ragged row tiles use the checked loaders.
  // TWO staging register sets: set (t & 1) carries tile t+2 into iteration t and is refilled with tile t+4 there, so
  // two k-tiles of global loads are in flight per workgroup (bytes in flight / latency is what bounds the global->LDS
  // rate: with one 16 KB tile in flight per 128x128 workgroup every K=512 shape sat at ~3.2 TB/s of loads = 100 TFLOP/s)
  f32x4 qreg[2][BM / 64], preg[2][BN / 64];
  // row-list dW GEMMs walk the reduce dimension through `kmap`: the indices of the NEXT staged k-tile are
  // fetched one iteration ahead (kq / kp), so the data loads never wait on an index load issued just before them
  int kq[BM / 64], kp[BN / 64];
  auto load_kidx = [&](int k0) {
    if constexpr (ROWS_ARE_K) {
      if (kmap) {
#pragma unroll
        for (int u = 0; u < BM / 64; ++u) kq[u] = kmap[min(k0 + (tid + 256 * u) / (BM / 4), Keff - 1)];
#pragma unroll
        for (int u = 0; u < BN / 64; ++u) kp[u] = kmap[min(k0 + (tid + 256 * u) / (BN / 4), Keff - 1)];
      }
    }
  };
  auto load_fast = [&](auto set_c, int k0, bool idx_ready) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int u = 0; u < BM / 64; ++u) {
      if constexpr (QKC) {
        qreg[S][u] = *reinterpret_cast<const f32x4*>(qfast[u] + k0);
      } else {
        const int gk = k0 + (tid + 256 * u) / (BM / 4);
        const int pk = kmap ? (idx_ready ? kq[u] : kmap[gk]) : gk;
        qreg[S][u] = *reinterpret_cast<const f32x4*>(qfast[u] + (int64_t)pk * g.ldq);
      }
    }
#pragma unroll
    for (int u = 0; u < BN / 64; ++u) {
      if constexpr (PKC) {
        preg[S][u] = *reinterpret_cast<const f32x4*>(pfast[u] + k0);
      } else {
        const int gk = k0 + (tid + 256 * u) / (BN / 4);
        const int pk = kmap ? (idx_ready ? kp[u] : kmap[gk]) : gk;
        preg[S][u] = *reinterpret_cast<const f32x4*>(pfast[u] + (int64_t)pk * g.ldp);
      }
    }
  };
  auto load_tiles = [&](auto set_c, int k0) {
    constexpr int S = decltype(set_c)::value;
    if (VEC && q_full && p_full && (k0 + BK <= kend)) {
      load_fast(set_c, k0, false);
    } else {
      if constexpr (QKC) tile_load_kc<BM, VEC>(qreg[S], g.Q, g.ldq, qrow, k0, kend, tid);
      else tile_load_mc<BM, VEC>(qreg[S], g.Q, g.ldq, m0, Meff, k0, kend, kmap, tid);
      if constexpr (PKC) tile_load_kc<BN, VEC>(preg[S], g.P, g.ldp, prow, k0, kend, tid);
      else tile_load_mc<BN, VEC>(preg[S], g.P, g.ldp, n0, g.N, k0, kend, kmap, tid);
    }
  };
  const bool do_colsum = ROWS_ARE_K && (g.colsum_part || g.colsum_out) && tile_n == 0;
  f32x4 qsum = {0.f, 0.f, 0.f, 0.f};   // running row sums of the staged Q vectors (every k-tile is stored exactly once)
  auto store_tiles = [&](auto set_c, float* buf) {
    constexpr int S = decltype(set_c)::value;
    tile_store<BM, QKC>(buf, qreg[S], tid);
    tile_store<BN, PKC>(buf + QSZ, preg[S], tid);
    if constexpr (ROWS_ARE_K) {
      if (do_colsum) {
#pragma unroll
        for (int u = 0; u < BM / 64; ++u) qsum += qreg[S][u];
      }
    }
  };

  // ---- software pipeline.  Stages of k-tile t:  G(t) global->regs, W(t) regs->LDS[t&1], R(t) LDS->fragments,
  // C(t) MFMAs.  Iteration t runs C(t) while it issues R(t+1), W(t+2) and G(t+3): the fragment registers are the
  // third stage, so two LDS images are enough, there is ONE barrier per k-tile and nothing after it waits on
  // memory - the next iteration's first MFMAs have their operands in registers already.
  //   * W(t+2) overwrites the image of tile t, whose last read (R(t)) was issued in iteration t-1, before
  //     that iteration's barrier;  R(t+2) happens in iteration t+1, after this iteration's barrier.
  //   * fragments ROLL where the MFMA order allows it: a fragment register is reloaded for tile t+1 as soon as
  //     the last MFMA of tile t that reads it has been issued.  KC fragments hold the 4 k-steps of one 16-row
  //     tile, MC fragments hold one k-step of all the wave's tiles (frag_load_mc), so
  //        NT (KC, KC): groups by Q tile a -> qf[a] rolls, pf is double-buffered;
  //        NN (KC, MC) / TN (MC, MC): groups by k-step s -> the MC fragments roll, a KC Q is double-buffered.
  //     Inside a group the same accumulator comes back after >= 4 MFMAs (16 in the s-grouped order).
  constexpr bool GROUP_BY_S = !PKC;
  static_assert(PKC || TM == TN, "MC fragments of both operands assume square wave tiles");
  f32x4 qf[(QKC && GROUP_BY_S) ? 2 : 1][TM];   // KC Q: [a] = 4 k-steps of tile a
  f32x4 pf[2][TN];                              // KC P: [b] = 4 k-steps of tile b (double-buffered)
  fvec<TM> qv[4];                               // MC Q: [s] = tiles 0..TM-1 at k-step s
  fvec<TN> pv[4];                               // MC P: [s] = tiles 0..TN-1 at k-step s
  auto read_q_kc = [&](const float* buf, int set, int a) { qf[set][a] = frag_load_kc<BM>(buf, wm * WTM + a * 16, li, lg); };
  auto read_p_kc = [&](const float* buf, int set, int b) { pf[set][b] = frag_load_kc<BN>(buf + QSZ, wn * WTN + b * 16, li, lg); };
  auto read_q_mc = [&](const float* buf, int s) { qv[s] = frag_load_mc<BM, TM>(buf, wm * WTM, li, lg, s); };
  auto read_p_mc = [&](const float* buf, int s) { pv[s] = frag_load_mc<BN, TN>(buf + QSZ, wn * WTN, li, lg, s); };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (nk > 0) {
    load_tiles(S0{}, kbeg);
    store_tiles(S0{}, smem);
    if (nk > 1) load_tiles(S1{}, kbeg + BK);
  }
  __syncthreads();
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!QKC) read_q_mc(smem, i);
      if constexpr (!PKC) read_p_mc(smem, i);
    }
    if constexpr (QKC) {
#pragma unroll
      for (int a = 0; a < TM; ++a) read_q_kc(smem, 0, a);
    }
    if constexpr (PKC) {
#pragma unroll
      for (int b = 0; b < TN; ++b) read_p_kc(smem, 0, b);
    }
    if (nk > 1) {
      store_tiles(S1{}, smem + BUF);
      if (nk > 2) load_tiles(S0{}, kbeg + 2 * BK);
      if (nk > 3) load_tiles(S1{}, kbeg + 3 * BK);
    }
  }
  __syncthreads();

  NACF_TRACE_MARK(1);
#ifdef NACF_GEMM_TRACE
  if (g_gemm_trace && threadIdx.x == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gemm_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
    g_gemm_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + 5] = (unsigned long long)nk;
    g_gemm_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + 6] = wall_clock64();
  }
#endif
  // STEADY iterations are straight-line code (no guards: tile kt+3 exists and is an interior tile of fully
  // populated row tiles), so the scheduler is free to interleave the LDS / global traffic with the MFMAs.
  // The loop body holds both parities (the staging set index must be a compile-time constant); every load is
  // consumed in the NEXT trip (G(kt+4) -> W(kt+4) two iterations later), so the optimiser cannot sink it to its use.
  // Double-buffered fragments are still rotated with register moves.
  auto iteration = [&](auto steady, auto set_c, int kt) {     // set_c = kt & 1 (compile time): the staging set of this parity
    constexpr bool STEADY = decltype(steady)::value;
    const float* nbuf = smem + ((kt + 1) & 1) * BUF;   // image of tile kt+1
    float* wbuf = smem + (kt & 1) * BUF;               // image of tile kt, about to become tile kt+2
    const bool has1 = STEADY || kt + 1 < nk, has2 = STEADY || kt + 2 < nk, has4 = STEADY || kt + 4 < nk;
    auto stage_ahead = [&]() {   // W(kt+2) out of this parity's staging set, then G(kt+4) into it
      if (has2) store_tiles(set_c, wbuf);
      if constexpr (STEADY) {
        load_fast(set_c, kbeg + (kt + 4) * BK, true);
        load_kidx(kbeg + (kt + 5) * BK);
        // keep W(kt+2) / G(kt+4) up here, ahead of their consumers
        // (MFMA, VALU, SALU and LDS reads may still be scheduled across; VMEM and LDS writes may not)
        __builtin_amdgcn_sched_barrier(0x10E);
      } else if (has4) {
        load_tiles(set_c, kbeg + (kt + 4) * BK);
      }
    };
    if constexpr (!GROUP_BY_S) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[0][b][s], qf[0][a][s], acc[a][b], 0, 0, 0);
        if (a == 0) {
          if (has1) {
#pragma unroll
            for (int b = 0; b < TN; ++b) read_p_kc(nbuf, 1, b);
          }
          stage_ahead();
        }
        if (has1) read_q_kc(nbuf, 0, a);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) pf[0][b] = pf[1][b];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            float qa;
            if constexpr (QKC) qa = qf[0][a][s]; else qa = qv[s][a];
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[s][b], qa, acc[a][b], 0, 0, 0);
          }
        if (s == 0) {
          if constexpr (QKC) {
            if (has1) {
#pragma unroll
              for (int a = 0; a < TM; ++a) read_q_kc(nbuf, 1, a);
            }
          }
          stage_ahead();
        }
        if (has1) {
          read_p_mc(nbuf, s);
          if constexpr (!QKC) read_q_mc(nbuf, s);
        }
      }
      if constexpr (QKC) {
#pragma unroll
        for (int a = 0; a < TM; ++a) qf[0][a] = qf[1][a];
      }
    }
    __syncthreads();
  };
  {
    // number of leading iterations whose G(kt+4) is an unguarded interior tile (even: the loop handles both parities
    // per trip so that the staging set index is a compile-time constant)
    int n_steady = 0;
    if (VEC && q_full && p_full) n_steady = min(nk - 4, (kend - kbeg) / BK - 4);
    if (n_steady < 0) n_steady = 0;
    n_steady &= ~1;
    int kt = 0;
    if (n_steady > 0) load_kidx(kbeg + 4 * BK);
#pragma nounroll
    for (; kt < n_steady; kt += 2) {
      iteration(std::true_type{}, S0{}, kt);
      iteration(std::true_type{}, S1{}, kt + 1);
    }
#pragma nounroll
    for (; kt + 1 < nk; kt += 2) {
      iteration(std::false_type{}, S0{}, kt);
      iteration(std::false_type{}, S1{}, kt + 1);
    }
    if (kt < nk) iteration(std::false_type{}, S0{}, kt);
  }

  NACF_TRACE_MARK(2);
  if constexpr (ROWS_ARE_K) {
    if (do_colsum) {
      // thread (kk = q / (BM/4), r4 = q % (BM/4)) holds the sums of rows 4*r4 .. 4*r4+3 over its k rows: fold the
      // lanes of a wave that share r4, then the 4 waves through LDS (fixed order: deterministic)
      constexpr int RV = BM / 4;
#pragma unroll
      for (int o = 32; o >= RV && o >= 1; o >>= 1) {
        qsum[0] += __shfl_xor(qsum[0], o, 64); qsum[1] += __shfl_xor(qsum[1], o, 64);
        qsum[2] += __shfl_xor(qsum[2], o, 64); qsum[3] += __shfl_xor(qsum[3], o, 64);
      }
      float* red = smem;   // [4 waves][BM]; the tile images are dead after the last barrier of the main loop
      if (lane < RV && lane < 64) *reinterpret_cast<f32x4*>(&red[wave * BM + 4 * lane]) = qsum;
      __syncthreads();
      if (tid < BM) {
        const int m = m0 + tid;
        if (m < g.M) {
          const float t = ((red[tid] + red[BM + tid]) + red[2 * BM + tid]) + red[3 * BM + tid];
          if (g.colsum_out) g.colsum_out[m] = (g.colsum_beta != 0.f) ? t + g.colsum_beta * g.colsum_out[m] : t;
          else g.colsum_part[(int64_t)z * g.M + m] = t;
        }
      }
    }
  }

  // ---- accumulator map: which (m, n) this lane's acc[a][b][e] is.
  //   KC Q: row  = a*16 + li                  MC Q: row = TM*li + a            (inside the wave tile)
  //   KC P: col  = b*16 + lg*4 + e            MC P: col = TN*(lg*4 + e) + b
  // The epilogues want float4s of 4 consecutive columns: with a KC P that is acc[a][b] itself; with an MC P
  // the 4 consecutive columns are spread over the TN tiles (and, for TN = 2, two values of e): out_vec().
  int mlog[TM], ncol[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) mlog[a] = m0 + wm * WTM + (QKC ? a * 16 + li : TM * li + a);
#pragma unroll
  for (int j = 0; j < TN; ++j) ncol[j] = n0 + wn * WTN + (PKC ? j * 16 + lg * 4 : 4 * TN * lg + 4 * j);

  if constexpr (!Epi::kArgmax) {
    // compile-time recursion, NOT a `#pragma unroll` loop: with the large fused epilogue the
    // optimiser declined to unroll, indexed acc[][] dynamically and parked the accumulators
    // in scratch (16 scratch_store_dwordx4 per k-tile in the main loop, 2x slower)
    int mphys[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) mphys[a] = (g.rows && !ROWS_ARE_K && mlog[a] < Meff) ? g.rows[mlog[a]] : mlog[a];
    if (epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N) epi.template tile_fast<TM, TN, PKC>(acc, mphys, ncol, g.N, z);
    else epilogue_all<0, TM, TN, PKC, Epi>(epi, acc, mlog, mphys, ncol, Meff, g.N, z);
#ifdef NACF_GEMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    if (g_gemm_trace && threadIdx.x == 0)
      g_gemm_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + 7] = wall_clock64();
#endif
    NACF_TRACE_MARK(3);
  } else {
    argmax_epilogue<BM, BN, WM, WN, TM, TN>(smem, g, epi, acc, m0, n0, Meff, tile_n, wm, wn, li, lg, tid);
  }
}
