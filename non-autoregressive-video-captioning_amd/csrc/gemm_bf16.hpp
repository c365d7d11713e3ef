// GEMMs on the CDNA4 bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate) with the SAME operand
// conventions, row-set handling, split-K and fused epilogues as gemm_f32.hpp:
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]
//
// The fp32 MFMA of gemm_f32.hpp runs at the fp32 VECTOR rate (157 TFLOP/s); the bf16 MFMA is 16x faster per
// instruction.  Two ways of using it, selected per launch (template parameter NS = number of bf16 terms an fp32
// operand is split into):
//
//   NS = 1  "throughput": operands rounded to bf16 (RNE), fp32 accumulate.  One MFMA per 16x16x32 block.
//   NS = 3  "exact":      every fp32 operand x is split EXACTLY into three bf16 terms x = h + m + l
//                         (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); 3 x 8 significand bits = the 24 of fp32)
//                         and the product block is formed from the six largest cross terms
//                             Pl*Qh + Ph*Ql + Pm*Qm + Pm*Qh + Ph*Qm + Ph*Qh          (small terms first)
//                         accumulated in fp32.  The three dropped terms are <= 2^-24 of the product, i.e. at fp32
//                         round-off: measured error against fp64 is at or below the fp32-MFMA kernel's (tests), and
//                         greedy NA-decode tokens stay bit-exact against the reference.  6 MFMAs per block:
//                         2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-accurate product rate vs 157 on the fp32 MFMA.
//
// Operand sources (compile time, per operand):
//   SRC_F32_KC   fp32, reduce-dim contiguous  (activations x, dZ)          -> converted / split while staging
//   SRC_F32_MC   fp32, row-dim contiguous     (both operands of dW)        -> 4x4 register transpose + convert
//   SRC_BF16_KC  bf16 image(s) of a weight matrix, k-TILE-MAJOR: [k / 32][row][32]   -> copied (no VALU work).
//                The images (W for the forward GEMM, W^T for dX) are rebuilt from the fp32 master weights once
//                per step (nacf_wimage_refresh): weights are reused by every row tile and by forward AND backward,
//                so they are split once, not once per tile -- and in OUR layout, so that the 64 bytes a row
//                contributes to a k-tile sit next to its neighbours' (a wave's load covers 8 full 128-byte lines;
//                with a row-major image it touched 16 half-used ones and the k-loop was bound by the texture
//                addresser: 34 cycles per wave load).
//
// LDS always holds bf16, K-contiguous: a plane is [rows][32 k] = 64-byte rows, 16-byte chunks XOR-swizzled with
// lds_sw(row) exactly as the KC tiles of gemm_f32.hpp (same geometry: conflict-free ds_read_b128 fragments,
// conflict-free ds_write_b128 staging).  A fragment (row = lane & 15, k-chunk = lane >> 4) is ONE ds_read_b128
// per split plane and feeds one K = 32 MFMA.  The MFMA is issued swapped (a = P fragment, b = Q fragment) so the
// accumulator map equals the KC/KC map of gemm_f32.hpp and its epilogues (EpiLinear / EpiStore / EpiArgmax) are
// reused unchanged.
//
// Pipeline: global -> registers one k-tile ahead, registers -> LDS (convert / split here, once per element),
// LDS -> fragments -> MFMA.  STAGES = 2 LDS images (one barrier per k-tile) where they fit; the exact mode's
// 128x128 tile (48 KB per image) uses one image and two barriers -- a k-tile there is 96 MFMAs per wave, and two
// to three workgroups per CU cover each other's barriers.
#pragma once
#include <type_traits>
#include "gemm_f32.hpp"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// native vectors (HIP's u32x4 / u32x2 are class types: arrays of them inside the staging structs ended up in scratch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

enum : int { SRC_F32_KC = 0, SRC_F32_MC = 1, SRC_BF16_KC = 2 };

// Tuning aid, compiled in only with -DNACF_BF16_TRACE (make trace; tools/bf16_trace.py): wave 0 of every workgroup
// adds up the shader-clock cycles it spends in each phase of the k-loop and stores them at the end:
//   [0] compute (fragment reads + MFMA issue)  [1] barrier after compute  [2] wait for the staged global loads
//   [3] convert / split + LDS stores + issue of the next loads  [4] barrier after the stores  [5] whole kernel  [6] k-tiles
#ifdef NACF_BF16_TRACE
__device__ unsigned long long* g_bf16_trace = nullptr;
#define BF16_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define BF16_TACC(slot, a, b) tacc[slot] += (b) - (a)
#else
#define BF16_T(var) do { } while (0)
#define BF16_TACC(slot, a, b) do { } while (0)
#endif

// two fp32 -> packed bf16 pair (a in the low half), round-to-nearest-even: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t bf16_pack_rne(float a, float b) {
  bf16x2_t r = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo_f32(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_f32(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// (a, b) -> NS packed bf16 pairs (a in the low halves).
//   NS = 1: round to nearest even (v_cvt_pk_bf16_f32).
//   NS = 3: EXACT split by truncation, x = h + m + l: h = the top 16 bits of x (8 significant bits), m = the top 16 bits
//           of x - h, l = x - h - m.  Each residual is exact in fp32 and loses >= 8 leading significant bits, so after
//           two steps at most 8 remain and l is a bf16 value: the three terms add up to x bit for bit.  Only full-rate
//           integer / add ops (v_and, v_sub, v_perm): 11 per pair.
__device__ __forceinline__ uint32_t bf16_pack_top(float a, float b) {      // {top16(b), top16(a)}
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float f32_top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
template <int NS>
__device__ __forceinline__ void bf16_split2(float a, float b, uint32_t (&w)[NS]) {
  if constexpr (NS == 1) {
    w[0] = bf16_pack_rne(a, b);
  } else {
    static_assert(NS == 3, "1 or 3 terms");
    w[0] = bf16_pack_top(a, b);
    a -= f32_top16(a);
    b -= f32_top16(b);
    w[1] = bf16_pack_top(a, b);
    a -= f32_top16(a);
    b -= f32_top16(b);
    w[2] = bf16_pack_top(a, b);
  }
}

#ifndef NACF_X3_DEEP
#define NACF_X3_DEEP 1
#endif

// ---------------------------------------------------------------- staging: global -> registers -> LDS planes
// An LDS plane of R rows is R*4 16-byte chunks: chunk (row, c) at plane[row * 4 + (c ^ lds_sw(row))].

// fp32, k-contiguous rows: thread unit = (row, 16-byte chunk j of the row's 128-byte k-tile segment), 8 lanes per row,
// so every wave load covers 8 whole 128-byte lines; one float4 load and one ds_write_b64 per plane per unit
template <int R, int NS>
struct StageF32KC {
  static constexpr int U = R / 32;
  f32x4 v[U];
  const float* ptr[U];   // (clamped) physical row base + 4 * j
  bool ok[U];            // row inside the matrix
  __device__ __forceinline__ void init(const float* base, int64_t ld, int r0, int rmax, const int* rowlist, int tid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int gr = r0 + (tid >> 3) + 32 * u;
      const int gc = min(gr, rmax - 1);
      const int ph = rowlist ? rowlist[gc] : gc;
      ok[u] = gr < rmax;
      ptr[u] = base + (int64_t)ph * ld + (tid & 7) * 4;
    }
  }
  __device__ __forceinline__ void load_fast(int k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(ptr[u] + k0);
  }
  __device__ __forceinline__ void load_checked(int k0, int kend, int tid) {
    const int nv = kend - (k0 + 4 * (tid & 7));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (ok[u] && nv > 0) {
        const float* p = ptr[u] + k0;
        if (nv >= 4) t = *reinterpret_cast<const f32x4*>(p);
        else {
          t[0] = p[0];
          if (nv > 1) t[1] = p[1];
          if (nv > 2) t[2] = p[2];
        }
      }
      v[u] = t;
    }
  }
  // split() + store() = write(), in two steps so that the conversion can run one pipeline phase before the LDS stores
  u32x2 w[U][NS];
  __device__ __forceinline__ void split_unit(int u) {
    uint32_t wa[NS], wb[NS];
    bf16_split2<NS>(v[u][0], v[u][1], wa);
    bf16_split2<NS>(v[u][2], v[u][3], wb);
#pragma unroll
    for (int p = 0; p < NS; ++p) w[u][p] = u32x2{wa[p], wb[p]};
  }
  __device__ __forceinline__ void split() {
#pragma unroll
    for (int u = 0; u < U; ++u) split_unit(u);
  }
  // the same split in 7 * 2U single steps (1-2 vector instructions each) for hand-interleaving with matrix
  // instructions; IN PLACE: v is consumed.  pair = (unit, half), step 0..6; both fold to constants when unrolled
  static constexpr int PAIRS = 2 * U;
  __device__ __forceinline__ void split_step(int pair, int step) {
    if constexpr (NS == 3) {
      const int u = pair >> 1, h = pair & 1;
      float a = v[u][2 * h], b = v[u][2 * h + 1];
      if (step == 0) w[u][0][h] = bf16_pack_top(a, b);
      else if (step == 1 || step == 4) v[u][2 * h] = a - f32_top16(a);
      else if (step == 2 || step == 5) v[u][2 * h + 1] = b - f32_top16(b);
      else if (step == 3) w[u][1][h] = bf16_pack_top(a, b);
      else w[u][2][h] = bf16_pack_top(a, b);
    }
  }
  __device__ __forceinline__ void store(u32x4* plane0, int tid) const {
    u32x2* p8 = reinterpret_cast<u32x2*>(plane0);
    const int j = tid & 7;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = (tid >> 3) + 32 * u;
#pragma unroll
      for (int p = 0; p < NS; ++p) p8[(p * (R * 4) + row * 4 + ((j >> 1) ^ lds_sw(row))) * 2 + (j & 1)] = w[u][p];
    }
  }
  __device__ __forceinline__ void write(u32x4* plane0, int tid) {
    split();
    store(plane0, tid);
  }
};

// fp32, row-contiguous (the reduce index walks the matrix ROWS, optionally through `kmap`): thread unit = 4 rows x
// 4 k, four float4 loads (one per k), a 4x4 transpose in registers, one ds_write_b64 per row per plane.
// unit = (rq = row quad, mq = k quad) with mq fastest: a wave's 64 lanes read 8 k-rows x 128 contiguous bytes.
// ROT rotates the thread -> unit map by 128 threads (R = 64 has 128 units: the second operand goes to waves 2-3).
template <int R, int NS, int ROT>
struct StageF32MC {
  static constexpr int UNITS = R * 2;
  f32x4 v[4];
  int kidx[4];           // physical k rows of the NEXT tile to load (prefetched when a kmap is given)
  const float* ptr;      // base + r0 + 4 * rq
  int rq, mq, nvalid;
  bool active;
  __device__ __forceinline__ void init(const float* base, int r0, int rmax, int tid) {
    const int unit = (tid + ROT) & 255;
    active = unit < UNITS;
    mq = unit & 7;
    rq = unit >> 3;
    ptr = base + r0 + 4 * rq;
    nvalid = min(4, max(0, rmax - (r0 + 4 * rq)));
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = f32x4{0.f, 0.f, 0.f, 0.f}; kidx[j] = 0; }
  }
  __device__ __forceinline__ void load_kidx(int k0, int kmax, const int* __restrict__ kmap) {
    if (kmap && active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) kidx[j] = kmap[min(k0 + 4 * mq + j, kmax - 1)];
    }
  }
  // interior k-tile of a fully populated row tile; idx_ready: kidx holds this tile's rows
  __device__ __forceinline__ void load_fast(int k0, int64_t ld, const int* __restrict__ kmap, bool idx_ready) {
    if (active) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gk = k0 + 4 * mq + j;
        const int pk = kmap ? (idx_ready ? kidx[j] : kmap[gk]) : gk;
        v[j] = *reinterpret_cast<const f32x4*>(ptr + (int64_t)pk * ld);
      }
    }
  }
  __device__ __forceinline__ void load_checked(int k0, int kend, int64_t ld, const int* __restrict__ kmap) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gk = k0 + 4 * mq + j;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (active && gk < kend && nvalid > 0) {
        const int pk = kmap ? kmap[gk] : gk;
        const float* p = ptr + (int64_t)pk * ld;
        if (nvalid == 4) t = *reinterpret_cast<const f32x4*>(p);
        else {
          t[0] = p[0];
          if (nvalid > 1) t[1] = p[1];
          if (nvalid > 2) t[2] = p[2];
        }
      }
      v[j] = t;
    }
  }
  u32x2 w[4][NS];
  __device__ __forceinline__ void split() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t wa[NS], wb[NS];
      bf16_split2<NS>(v[0][e], v[1][e], wa);
      bf16_split2<NS>(v[2][e], v[3][e], wb);
#pragma unroll
      for (int p = 0; p < NS; ++p) w[e][p] = u32x2{wa[p], wb[p]};
    }
  }
  static constexpr int PAIRS = 8;
  __device__ __forceinline__ void split_step(int pair, int step) {     // see StageF32KC::split_step
    if constexpr (NS == 3) {
      const int e = pair >> 1, h = pair & 1;
      float a = v[2 * h][e], b = v[2 * h + 1][e];
      if (step == 0) w[e][0][h] = bf16_pack_top(a, b);
      else if (step == 1 || step == 4) v[2 * h][e] = a - f32_top16(a);
      else if (step == 2 || step == 5) v[2 * h + 1][e] = b - f32_top16(b);
      else if (step == 3) w[e][1][h] = bf16_pack_top(a, b);
      else w[e][2][h] = bf16_pack_top(a, b);
    }
  }
  __device__ __forceinline__ void store(u32x4* plane0) const {
    if (active) {
      u32x2* p8 = reinterpret_cast<u32x2*>(plane0);
      const int c = mq >> 1, half = mq & 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 4 * rq + e;
#pragma unroll
        for (int p = 0; p < NS; ++p) p8[(p * (R * 4) + row * 4 + (c ^ lds_sw(row))) * 2 + half] = w[e][p];
      }
    }
  }
  __device__ __forceinline__ void write(u32x4* plane0) {
    split();
    store(plane0);
  }
};

// bf16 image planes, k-tile-major ([k / 32][row][32], zero-padded to whole k-tiles): a row tile of one k-tile is
// R * 64 contiguous bytes; copied
template <int R, int NS>
struct StageBF16KC {
  static constexpr int U = R / 64;
  u32x4 v[U][NS];
  const unsigned short* ptr[U];   // k-tile 0: (clamped) row * 32 + 8 * chunk
  __device__ __forceinline__ void init(const unsigned short* img, int r0, int rmax, int tid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = tid + 256 * u;
      const int64_t rowoff = (int64_t)min(r0 + (q >> 2), rmax - 1) * 32;
      ptr[u] = img + rowoff + (q & 3) * 8;
    }
  }
  // kt_stride: elements between consecutive k-tiles (= rows of the registered matrix * 32)
  __device__ __forceinline__ void load(int k0, int64_t kt_stride, int64_t plane) {
    const int64_t off = (int64_t)(k0 >> 5) * kt_stride;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < NS; ++p) v[u][p] = *reinterpret_cast<const u32x4*>(ptr[u] + p * plane + off);
  }
  __device__ __forceinline__ void write(u32x4* plane0, int tid) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = tid + 256 * u;
      const int row = q >> 2, c = q & 3;
#pragma unroll
      for (int p = 0; p < NS; ++p) plane0[p * (R * 4) + row * 4 + (c ^ lds_sw(row))] = v[u][p];
    }
  }
};

template <int SRC, int R, int NS, int ROT> struct StagerOf;
template <int R, int NS, int ROT> struct StagerOf<SRC_F32_KC, R, NS, ROT> { using type = StageF32KC<R, NS>; };
template <int R, int NS, int ROT> struct StagerOf<SRC_F32_MC, R, NS, ROT> { using type = StageF32MC<R, NS, ROT>; };
template <int R, int NS, int ROT> struct StagerOf<SRC_BF16_KC, R, NS, ROT> { using type = StageBF16KC<R, NS>; };

__device__ __forceinline__ bf16x8_t lds_frag(const u32x4* plane, int row, int g) {
  return __builtin_bit_cast(bf16x8_t, plane[row * 4 + (g ^ lds_sw(row))]);
}

// dynamic LDS of a launch in 16-byte chunks (tile images, or the epilogue's reduction scratch if that is larger)
template <int BM, int BN, int NS, int STAGES, bool ARGMAX>
constexpr int gemm_bf16_lds_chunks() {
  const int qst = (STAGES == 2) ? 2 : 1, pst = (STAGES == 1) ? 1 : 2;
  const int tiles = NS * (qst * BM * 4 + pst * BN * 4);
  const int red = ARGMAX ? (3 * 2 * BM + 3) / 4 : (BM + 3) / 4;
  return tiles > red ? tiles : red;
}

// ---------------------------------------------------------------- kernel
// The body of one workgroup: output tile `bid` (XCD-remapped below) of reduce split `z` out of `nz`.  Shared by the plain
// launch (one problem per grid: bid = blockIdx.x, z = blockIdx.z) and the grouped launch (a table of problems per grid).
template <int BM, int BN, int QSRC, int PSRC, int NS, int STAGES, class Epi>
__device__ __forceinline__ void gemm_bf16_body(const GemmShape& g, const Epi& epi, const int bid, const int z, const int nz,
                                               const bool mapped = false) {
  constexpr int BK = 32;
  constexpr int WM = 2, WN = 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr bool QKC = QSRC != SRC_F32_MC, PKC = PSRC != SRC_F32_MC;
  constexpr bool ROWS_ARE_K = !QKC && !PKC;
  static_assert(QSRC != SRC_BF16_KC, "the Q operand is an activation (fp32)");
  static_assert(QKC || !PKC, "layouts: NT (KC, KC), NN via a transposed image (KC, KC), NN (KC, MC), TN (MC, MC)");
  static_assert(NS == 1 || NS == 3, "1 = bf16 throughput, 3 = exact fp32 split");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "stager granularity");
  constexpr int QPL = BM * 4, PPL = BN * 4;            // 16-byte chunks per plane
  // STAGES: 1 = one LDS image of both operands (two barriers per k-tile); 2 = two images (one barrier);
  //         3 = one image of Q, TWO of P: the P loads of tile t+2 are issued at the start of the MFMA phase of tile t
  //             (their address processing overlaps the matrix work) and stored during the staging phase of tile t
  constexpr int QST = (STAGES == 2) ? 2 : 1, PST = (STAGES == 1) ? 1 : 2;
  constexpr int QBASE = 0, PBASE = QST * NS * QPL;
  static_assert(gemm_bf16_lds_chunks<BM, BN, NS, STAGES, Epi::kArgmax>() >= PBASE + PST * NS * PPL, "LDS size helper out of date");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smem = reinterpret_cast<u32x4*>(smem_raw);
  auto q_stage = [&](int st) { return smem + QBASE + st * (NS * QPL); };
  auto p_stage = [&](int st) { return smem + PBASE + st * (NS * PPL); };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;

  // live-row list: NT / NN bound the row walk, TN bounds the reduce walk (see gemm_f32.hpp)
  int Meff = g.M, Keff = g.K;
  if (g.count) {
    const int c = *g.count;
    if (ROWS_ARE_K) Keff = min(Keff, c);
    else Meff = min(Meff, c);
  }
  const int tiles_m_live = (Meff + BM - 1) / BM;
  const int nwg = tiles_m_live * g.tiles_n;
  if (bid >= nwg) {
    // workgroups past the live tiles zero-fill the dead rows of the output (nacf_rowset.zero_dead)
    if constexpr (!ROWS_ARE_K) {
      if (g.zero_dead && g.rows && z == 0) {
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
  // XCD-aware bijective tile order over the LIVE tiles (n fastest inside an XCD's contiguous run)
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  // (mapped: the caller has already undone the round-robin of workgroups over XCDs -- `bid` IS the tile)
  const int logical = mapped ? bid : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
  int tile_m = logical / g.tiles_n, tile_n = logical % g.tiles_n;
  if (g.group_n > 0) {
    const int per = tiles_m_live * g.group_n;
    const int grp = logical / per, r = logical - grp * per;
    const int gn = min(g.group_n, g.tiles_n - grp * g.group_n);
    tile_m = r / gn;
    tile_n = grp * g.group_n + (r - tile_m * gn);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int kps = g.k_per_split;
  if (ROWS_ARE_K && g.count && nz > 1) {
    kps = (((Keff + nz - 1) / nz) + BK - 1) / BK * BK;   // re-balance over the LIVE rows
    if (kps < BK) kps = BK;
  }
  const int kbeg = z * kps;
  const int kend = min(Keff, kbeg + kps);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int* kmap = ROWS_ARE_K ? g.rows : nullptr;

  typename StagerOf<QSRC, BM, NS, 0>::type qs;
  typename StagerOf<PSRC, BN, NS, (BN == 64 ? 128 : 0)>::type ps;
  if constexpr (QSRC == SRC_F32_KC) qs.init(g.Q, g.ldq, m0, Meff, g.rows, tid);
  else qs.init(g.Q, m0, g.M, tid);
  if constexpr (PSRC == SRC_F32_KC) ps.init(g.P, g.ldp, n0, g.N, nullptr, tid);
  else if constexpr (PSRC == SRC_F32_MC) ps.init(g.P, n0, g.N, tid);
  else ps.init(g.Pimg, n0, g.N, tid);
  // row-contiguous operands take the unchecked loader only when the whole row tile is inside the matrix
  const bool q_full = QKC || (m0 + BM <= g.M);
  const bool p_full = PKC || (n0 + BN <= g.N);
  const bool rows_full = q_full && p_full;

  auto load_q = [&](int kt, bool idx_ready, bool steady = false) {
    const int k0 = kbeg + kt * BK;
    const bool fast = steady || (rows_full && (k0 + BK <= kend));
    if constexpr (QSRC == SRC_F32_KC) { if (fast) qs.load_fast(k0); else qs.load_checked(k0, kend, tid); }
    else { if (fast) qs.load_fast(k0, g.ldq, kmap, idx_ready); else qs.load_checked(k0, kend, g.ldq, kmap); }
    if constexpr (ROWS_ARE_K) { if (kmap && kt + 1 < nk) qs.load_kidx(kbeg + (kt + 1) * BK, Keff, kmap); }   // next tile's rows
  };
  auto load_p = [&](int kt, bool idx_ready, bool steady = false) {
    const int k0 = kbeg + kt * BK;
    const bool fast = steady || (rows_full && (k0 + BK <= kend));
    if constexpr (PSRC == SRC_F32_KC) { if (fast) ps.load_fast(k0); else ps.load_checked(k0, kend, tid); }
    else if constexpr (PSRC == SRC_F32_MC) {
      if (fast) ps.load_fast(k0, g.ldp, kmap, idx_ready); else ps.load_checked(k0, kend, g.ldp, kmap);
      if constexpr (ROWS_ARE_K) { if (kmap && kt + 1 < nk) ps.load_kidx(kbeg + (kt + 1) * BK, Keff, kmap); }
    }
    else ps.load(k0, g.ldpi, g.pimg_plane);      // zero-padded to whole k-tiles, rows clamped: never out of bounds
  };
  const bool do_colsum = ROWS_ARE_K && (g.colsum_part || g.colsum_out) && tile_n == 0;
  f32x4 qsum = {0.f, 0.f, 0.f, 0.f};
  auto write_q = [&](int st) {
    if constexpr (QSRC == SRC_F32_KC) qs.write(q_stage(st), tid);
    else {
      qs.write(q_stage(st));
      if (do_colsum) qsum += (qs.v[0] + qs.v[1]) + (qs.v[2] + qs.v[3]);   // every k-tile is written exactly once
    }
  };
  auto store_q = [&](int st) {          // second half of write_q (the STAGES == 3 steady loop splits in compute_split)
    if constexpr (QSRC == SRC_F32_KC) qs.store(q_stage(st), tid);
    else qs.store(q_stage(st));
  };
  auto write_p = [&](int st) {
    if constexpr (PSRC == SRC_F32_MC) ps.write(p_stage(st));
    else ps.write(p_stage(st), tid);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int qst, int pst) {
    const u32x4* qpl = q_stage(qst);
    const u32x4* ppl = p_stage(pst);
    bf16x8_t pf[TN][NS];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int p = 0; p < NS; ++p) pf[b][p] = lds_frag(ppl + p * PPL, wn * WTN + b * 16 + li, lg);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      bf16x8_t qf[NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) qf[p] = lds_frag(qpl + p * QPL, wm * WTM + a * 16 + li, lg);
      if constexpr (NS == 1) {
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][0], qf[0], acc[a][b], 0, 0, 0);
      } else {
        // six cross terms, smallest first; the same accumulator comes back after TN MFMAs
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][2], qf[0], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][0], qf[2], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][1], qf[1], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][1], qf[0], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][0], qf[1], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][0], qf[0], acc[a][b], 0, 0, 0);
      }
    }
  };

  // C(kt) with the split of the Q registers (tile kt+1) threaded through it BY HAND: after every matrix instruction
  // one split step (1-2 full-rate vector instructions), pinned there by a scheduling fence -- the matrix pipe takes 16
  // cycles per instruction and the issue port is free for 12 of them, so the conversion costs no time of its own.
  // The Q fragments of row block a+1 are fetched a block ahead (the fences stop the compiler from hoisting them).
  // q_slot0: matrix-instruction slot of the first Q split step; kt_next >= 0: request that Q tile once the registers are split
  auto compute_split = [&](int pst, int kt_next, auto q_slot0_c) {
    constexpr int Q0 = decltype(q_slot0_c)::value;
    static_assert(NS == 3 || STAGES != 3, "hand schedule is for the three-term kernels");
    if constexpr (NS == 3 && QSRC != SRC_BF16_KC) {
      if constexpr (QSRC != SRC_F32_KC) { if (do_colsum) qsum += (qs.v[0] + qs.v[1]) + (qs.v[2] + qs.v[3]); }
      const u32x4* qpl = q_stage(0);
      const u32x4* ppl = p_stage(pst);
      bf16x8_t pf[TN][NS], qf[2][NS];
      // fragment reads in the order the terms need them (LDS returns data in order): the first matrix instruction waits
      // for 5 of the 15 reads instead of for 13 -- all four waves start their phase with 18 KB of reads each
      constexpr int PORD[3] = {2, 0, 1}, QORD[3] = {0, 2, 1};
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        qf[0][QORD[o]] = lds_frag(qpl + QORD[o] * QPL, wm * WTM + li, lg);
#pragma unroll
        for (int b = 0; b < TN; ++b) pf[b][PORD[o]] = lds_frag(ppl + PORD[o] * PPL, wn * WTN + b * 16 + li, lg);
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int NSTEP = 7 * decltype(qs)::PAIRS;
      constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};     // six cross terms, smallest first
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        if (a + 1 < TM) {
#pragma unroll
          for (int p = 0; p < NS; ++p) qf[(a + 1) & 1][p] = lds_frag(qpl + p * QPL, wm * WTM + (a + 1) * 16 + li, lg);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][TP[t]], qf[a & 1][TQ[t]], acc[a][b], 0, 0, 0);
            constexpr int NSLOT = TM * 6 * TN;
            const int slot = (a * 6 + t) * TN + b;
            static_assert(Q0 + NSTEP <= NSLOT, "split steps past the last matrix instruction");
            if (slot >= Q0 && slot < Q0 + NSTEP) qs.split_step((slot - Q0) / 7, (slot - Q0) % 7);
            if (slot == Q0 + NSTEP || (Q0 + NSTEP == NSLOT && slot == NSLOT - 1)) { if (kt_next >= 0) load_q(kt_next, true, true); }
            if constexpr (PSRC == SRC_F32_MC) {
              // a row-contiguous fp32 P (dW): its registers hold tile kt+2, split in the LAST slots; tile kt+3 is
              // requested after the loop
              constexpr int PSTEP = 7 * decltype(ps)::PAIRS;
              const int ps_slot = slot - (NSLOT - PSTEP);
              if (ps_slot >= 0) ps.split_step(ps_slot / 7, ps_slot % 7);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  };

  // the same for the two-image pipelines (STAGES == 2, the 64 x 64 exact tile): everything of tile kt+1 -- P stores, the
  // Q split and its stores, the loads of tile kt+2 -- shares one straight-line block with the MFMAs of tile kt.  Term
  // outermost: an accumulator comes back after TM * TN matrix instructions (the per-accumulator order of the six terms,
  // and with it every bit of the result, is that of compute()).
  // QS / PS: the register set that holds tile kt+1; `ahead` = how many tiles further the set is re-loaded for (2: the one
  // register set of the general pipeline; 3: two sets alternate, every load has TWO iterations to land -- see the loop)
  auto compute_split2 = [&](int kt, auto& QS, auto& PS, int ahead) {
    if constexpr (NS == 3 && STAGES == 2 && QSRC == SRC_F32_KC && PSRC == SRC_BF16_KC) {
      const int cur = kt & 1, nxt = cur ^ 1;
      PS.write(p_stage(nxt), tid);
      PS.load(kbeg + (kt + ahead) * BK, g.ldpi, g.pimg_plane);
      const u32x4* qpl = q_stage(cur);
      const u32x4* ppl = p_stage(cur);
      bf16x8_t pf[TN][NS], qf[TM][NS];
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int p = 0; p < NS; ++p) pf[b][p] = lds_frag(ppl + p * PPL, wn * WTN + b * 16 + li, lg);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int p = 0; p < NS; ++p) qf[a][p] = lds_frag(qpl + p * QPL, wm * WTM + a * 16 + li, lg);
      constexpr int NSTEP = 7 * decltype(qs)::PAIRS, NSLOT = 6 * TM * TN;
      constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][TP[t]], qf[a][TQ[t]], acc[a][b], 0, 0, 0);
            const int slot = (t * TM + a) * TN + b;
#pragma unroll
            for (int i = 0; i < NSTEP; ++i)
              if (i * NSLOT / NSTEP == slot) QS.split_step(i / 7, i % 7);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      QS.store(q_stage(nxt), tid);
      QS.load_fast(kbeg + (kt + ahead) * BK);
    }
  };

#ifdef NACF_BF16_TRACE
  unsigned long long tacc[5] = {0, 0, 0, 0, 0};
  unsigned long long tsub[3] = {0, 0, 0};     // STAGES == 3 staging phase: Q split + stores | Q load issue | P stores
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  // ---- pipeline: G(t) global -> registers, W(t) registers -> LDS image (convert / split), C(t) fragments + MFMAs
  if constexpr (ROWS_ARE_K) {
    if (kmap && nk > 0) { qs.load_kidx(kbeg, Keff, kmap); ps.load_kidx(kbeg, Keff, kmap); }
  }
  if (nk > 0) {
    load_q(0, true);
    load_p(0, true);
    write_q(0);
    write_p(0);
    if (nk > 1) {
      load_q(1, true);
      load_p(1, true);
      if constexpr (STAGES == 3) write_p(1);      // P runs two tiles ahead: tile 1 is stored now, tile 2 loaded in C(0)
    }
  }
  __syncthreads();
  int kt_first = 0;
  bool p_preloaded = false;      // the steady loop left tile kt_first+2 of P in the registers
#ifndef NACF_BF16_TRACE
  if constexpr (STAGES == 2) {
    // STEADY iterations (tile kt+2 exists and is an interior tile of fully populated row tiles): one straight-line
    // block -- stores of tile kt+1, unguarded loads of tile kt+2, fragments + MFMAs of tile kt -- so that the scheduler
    // may slide the staging instructions of the OTHER LDS image into the shadows of this tile's MFMAs.  A launch of
    // less than one round of workgroups (the decoder's 512-wide GEMMs) has nobody else to hide them behind.
    const int n_steady = rows_full ? min(nk - 2, (kend - kbeg) / BK - 2) : 0;
    if constexpr (NS == 3 && BM == 64 && QSRC == SRC_F32_KC && PSRC == SRC_BF16_KC && NACF_X3_DEEP) {
      // The 64 x 64 launches of the decoder layers are under one round of workgroups (<= 3 per CU): nothing hides a
      // global load but the workgroup's own next iteration, and an iteration (24 MFMAs per wave) is shorter than the
      // load latency under load -- the k-loop ran at one memory latency per k-tile.  Two register sets, alternating:
      // iteration kt consumes tile kt+1 from one set and re-loads that set with tile kt+3, so every load has two
      // iterations to land.  Unrolled by two (the sets are distinct variables); an even count keeps tile kt+1 in qs / ps
      // for the general loop that follows.
      int n_deep = rows_full ? min(nk - 3, (kend - kbeg) / BK - 3) : 0;
      n_deep &= ~1;
      if (n_deep > 0) {
        auto qs2 = qs;
        auto ps2 = ps;
        qs2.load_fast(kbeg + 2 * BK);
        ps2.load(kbeg + 2 * BK, g.ldpi, g.pimg_plane);
#pragma nounroll
        for (; kt_first < n_deep; kt_first += 2) {
          compute_split2(kt_first, qs, ps, 3);
          __syncthreads();
          compute_split2(kt_first + 1, qs2, ps2, 3);
          __syncthreads();
        }
      }
    }
#pragma nounroll
    for (; kt_first < n_steady; ++kt_first) {
      if constexpr (NS == 3 && BM == 64 && QSRC == SRC_F32_KC && PSRC == SRC_BF16_KC) {   // (dW's two converted operands measure slower this way)
        compute_split2(kt_first, qs, ps, 2);
      } else {
        write_q((kt_first + 1) & 1);
        write_p((kt_first + 1) & 1);
        load_q(kt_first + 2, true, true);
        load_p(kt_first + 2, true, true);
        compute(kt_first & 1, kt_first & 1);
      }
      __syncthreads();
    }
  }
#endif
  if constexpr (STAGES == 3 && NS == 3) {
    // STEADY iterations of the one-Q-image pipeline.  Invariant at the top of iteration kt: the Q image holds tile kt,
    // the P images tiles kt and kt+1, the Q registers the raw fp32 of tile kt+1 (and, for a converted P, the P
    // registers the raw tile kt+2).  The split of those registers into bf16 planes is pure VALU work with no LDS
    // dependence: compute_split threads it through the MFMAs of tile kt; only the stores wait for the barrier.  The
    // raw registers are free as soon as they are split, so the next loads go out most of an MFMA phase earlier too.
    constexpr bool PSHADOW = (PSRC == SRC_F32_MC);
    constexpr int AHEAD = PSHADOW ? 3 : 2;
    const int n_steady = rows_full ? min(nk - AHEAD, (kend - kbeg) / BK - AHEAD) : 0;
    if constexpr (PSHADOW) {
      if (n_steady > 0) { load_p(2, true, true); p_preloaded = true; }
    }
#pragma nounroll
    for (; kt_first < n_steady; ++kt_first) {
      BF16_T(t0);
      if constexpr (!PSHADOW) load_p(kt_first + 2, true, true);
      compute_split(kt_first & 1, kt_first + 2, std::integral_constant<int, 0>{});
      if constexpr (PSHADOW) load_p(kt_first + 3, true, true);
      BF16_T(t1);
      __syncthreads();
      BF16_T(t2);
      store_q(0);
      if constexpr (PSHADOW) ps.store(p_stage(kt_first & 1));
      else write_p(kt_first & 1);
      BF16_T(t4);
      __syncthreads();
      BF16_T(t5);
      BF16_TACC(0, t0, t1); BF16_TACC(1, t1, t2); BF16_TACC(3, t2, t4); BF16_TACC(4, t4, t5);
    }
  }
#pragma nounroll
  for (int kt = kt_first; kt < nk; ++kt) {
    if constexpr (STAGES == 2) {
      // W(kt+1) goes into the image C(kt-1) read (a barrier ago); G(kt+2) has all of C(kt) to land
      BF16_T(t0);
#ifdef NACF_BF16_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      BF16_T(t1);
      if (kt + 1 < nk) { write_q((kt + 1) & 1); write_p((kt + 1) & 1); }
      if (kt + 2 < nk) { load_q(kt + 2, true); load_p(kt + 2, true); }
      BF16_T(t2);
      compute(kt & 1, kt & 1);
      BF16_T(t3);
      __syncthreads();
      BF16_T(t4);
      BF16_TACC(2, t0, t1); BF16_TACC(3, t1, t2); BF16_TACC(0, t2, t3); BF16_TACC(1, t3, t4);
    } else if constexpr (STAGES == 1) {
      BF16_T(t0);
      compute(0, 0);
      BF16_T(t1);
      __syncthreads();
      BF16_T(t2);
#ifdef NACF_BF16_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      BF16_T(t3);
      if (kt + 1 < nk) {
        write_q(0);
        write_p(0);
        if (kt + 2 < nk) { load_q(kt + 2, true); load_p(kt + 2, true); }
      }
      BF16_T(t4);
      __syncthreads();
      BF16_T(t5);
      BF16_TACC(0, t0, t1); BF16_TACC(1, t1, t2); BF16_TACC(2, t2, t3); BF16_TACC(3, t3, t4); BF16_TACC(4, t4, t5);
    } else {
      // one Q image, two P images.  The P registers are free during C(kt) (tile kt+1 sits in LDS already), so the
      // loads of tile kt+2 are issued FIRST: the texture addresser works through them while the MFMAs run, and the
      // staging phase only converts / stores (plus the four Q loads)
      BF16_T(t0);
      if (kt + 2 < nk && !(p_preloaded && kt == kt_first)) load_p(kt + 2, true);
      compute(0, kt & 1);
      BF16_T(t1);
      __syncthreads();
      BF16_T(t2);
#ifdef NACF_BF16_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      BF16_T(t3);
      if (kt + 1 < nk) {
        write_q(0);
#ifdef NACF_BF16_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        BF16_T(t3a);
        if (kt + 2 < nk) {
          load_q(kt + 2, true);
          BF16_T(t3b);
          write_p(kt & 1);             // tile kt+2 into the image C(kt) has just released
#ifdef NACF_BF16_TRACE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long t3c = __builtin_readcyclecounter();
          tsub[0] += t3a - t3; tsub[1] += t3b - t3a; tsub[2] += t3c - t3b;
#endif
        }
      }
      BF16_T(t4);
      __syncthreads();
      BF16_T(t5);
      BF16_TACC(0, t0, t1); BF16_TACC(1, t1, t2); BF16_TACC(2, t2, t3); BF16_TACC(3, t3, t4); BF16_TACC(4, t4, t5);
    }
  }
#ifdef NACF_BF16_TRACE
  if (g_bf16_trace && tid == 0) {
    unsigned long long* o = g_bf16_trace + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;
    for (int i = 0; i < 5; ++i) o[i] = tacc[i];
    const unsigned long long t_loop_end = __builtin_readcyclecounter();
    o[5] = t_loop_end - t_begin;
    o[6] = (unsigned long long)nk;
    o[7] = (tsub[0] << 42) | (tsub[1] << 21) | tsub[2];      // three 21-bit sums (cycles / 16 would overflow less; fine for nk <= 64)
    o[8] = t_begin;                                           // absolute stamps: the timeline of workgroups per CU
    o[9] = t_loop_end;
    o[11] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID | HW_ID
  }
#endif

  if constexpr (ROWS_ARE_K) {
    if (do_colsum) {
      // db[n] = sum_m dZ[m][n]: the thread of unit (rq, mq) holds the sums of rows 4rq..4rq+3 over its k quads;
      // the 8 lanes of a row quad sit in one wave, so a 3-step butterfly finishes it (fixed order: deterministic)
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        qsum[0] += __shfl_xor(qsum[0], o, 64); qsum[1] += __shfl_xor(qsum[1], o, 64);
        qsum[2] += __shfl_xor(qsum[2], o, 64); qsum[3] += __shfl_xor(qsum[3], o, 64);
      }
      float* red = reinterpret_cast<float*>(smem);   // [BM]; the tile images are dead after the loop's last barrier
      if (qs.active && qs.mq == 0) *reinterpret_cast<f32x4*>(&red[4 * qs.rq]) = qsum;
      __syncthreads();
      if (tid < BM) {
        const int m = m0 + tid;
        if (m < g.M) {
          const float t = red[tid];
          if (g.colsum_out) g.colsum_out[m] = (g.colsum_beta != 0.f) ? t + g.colsum_beta * g.colsum_out[m] : t;
          else g.colsum_part[(int64_t)z * g.M + m] = t;
        }
      }
    }
  }

  // ---- accumulator map (= the KC / KC map of gemm_f32.hpp): acc[a][b][e] is row a*16 + li, column b*16 + lg*4 + e
  int mlog[TM], ncol[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) mlog[a] = m0 + wm * WTM + a * 16 + li;
#pragma unroll
  for (int j = 0; j < TN; ++j) ncol[j] = n0 + wn * WTN + j * 16 + lg * 4;
  if constexpr (!Epi::kArgmax) {
    int mphys[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) mphys[a] = (g.rows && !ROWS_ARE_K && mlog[a] < Meff) ? g.rows[mlog[a]] : mlog[a];
    if (epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N) epi.template tile_fast<TM, TN, true>(acc, mphys, ncol, g.N, z);
    else epilogue_all<0, TM, TN, true, Epi>(epi, acc, mlog, mphys, ncol, Meff, g.N, z);
  } else {
    argmax_epilogue<BM, BN, WM, WN, TM, TN>(reinterpret_cast<float*>(smem), g, epi, acc, m0, n0, Meff, tile_n, wm, wn, li, lg, tid);
  }
#ifdef NACF_BF16_TRACE
  if (g_bf16_trace && tid == 0) {
    const unsigned long long t_issued = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the epilogue's stores have left this wave
    g_bf16_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 10] = __builtin_readcyclecounter();
    g_bf16_trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 12] = t_issued;
  }
#endif
}

template <int BM, int BN, int QSRC, int PSRC, int NS, int STAGES, class Epi>
__global__ __launch_bounds__(256, (BM == 64 ? 3 : 2)) void gemm_bf16_kernel(GemmShape g, Epi epi) {
  gemm_bf16_body<BM, BN, QSRC, PSRC, NS, STAGES, Epi>(g, epi, (int)blockIdx.x, (int)blockIdx.z, (int)gridDim.z);
}

// ---------------------------------------------------------------- grouped launch
// Several INDEPENDENT problems of one kernel type in a single grid: workgroup range [wg0[p], wg0[p + 1]) belongs to
// problem p, laid out as nz[p] reduce splits of gx[p] workgroups each (gx a multiple of 8, so that the XCD a workgroup
// lands on -- blockIdx.x & 7 -- is the one the body's tile remap assumes; surplus workgroups return at once).  Built for
// the weight-gradient GEMMs of a backward pass: launched one at a time each has to split its reduce dimension 8-17 ways
// to fill 256 CUs; launched together 1-4 ways do, with k-loops that much longer and that many fewer slabs to combine.
//   order 1 (split-major, the weight-gradient default): the nz[p] x gx[p] (split, tile) pairs of a problem form ONE list,
//   tile fastest, and each XCD takes a contiguous eighth of it -- the workgroups that share an XCD's L2 then walk the SAME
//   reduce range, so a dZ / X row panel is fetched from HBM by about one XCD instead of by all eight (order 0: every
//   split spreads its tiles over the eight XCDs; measured 1.92 GB moved for 0.63 GB of operands).  gx[p] is then the
//   exact tile count and the problem owns ceil(nz * gx / 8) * 8 workgroups.
constexpr int GEMM_GROUP_MAX = 16;
template <class Epi>
struct GemmGroup {
  int n;
  int order;
  int wg0[GEMM_GROUP_MAX + 1];
  int gx[GEMM_GROUP_MAX];
  int nz[GEMM_GROUP_MAX];
  GemmShape g[GEMM_GROUP_MAX];
  Epi e[GEMM_GROUP_MAX];
};
template <int BM, int BN, int QSRC, int PSRC, int NS, int STAGES, class Epi>
__global__ __launch_bounds__(256, (BM == 64 ? 3 : 2)) void gemm_bf16_group_kernel(GemmGroup<Epi> t) {
  int p = 0;
#pragma unroll 1
  while (p + 1 < t.n && (int)blockIdx.x >= t.wg0[p + 1]) ++p;
  const int local = (int)blockIdx.x - t.wg0[p];
  const int gx = t.gx[p];
  if (t.order == 1) {
    const int total = gx * t.nz[p];
    const int xq = total >> 3, xr = total & 7;
    const int xcd = local & 7, slot = local >> 3;
    if (slot >= xq + (xcd < xr ? 1 : 0)) return;
    const int l = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
    const int z = l / gx;
    gemm_bf16_body<BM, BN, QSRC, PSRC, NS, STAGES, Epi>(t.g[p], t.e[p], l - z * gx, z, t.nz[p], true);
    return;
  }
  const int z = local / gx;
  gemm_bf16_body<BM, BN, QSRC, PSRC, NS, STAGES, Epi>(t.g[p], t.e[p], local - z * gx, z, t.nz[p]);
}

// ---------------------------------------------------------------- weight images
// One launch rebuilds every registered image from the fp32 master weights: for matrix i (N x K, row pitch ld),
//   img [s][(k / 32) * N + n][k % 32] = bf16 term s of W[n][k]     (forward: P = W,   reduce over k, zero-padded to 32)
//   imgT[s][(n / 32) * K + k][n % 32] = bf16 term s of W[n][k]     (dX:      P = W^T, reduce over n, zero-padded to 32)
// A workgroup converts one 32 x 32 tile; the transposed copies go through LDS so that every store is at least 8 bytes.
typedef nacf_wimage_desc WImageDesc;   // include/nacf_hip.h

template <int NS>
__global__ __launch_bounds__(256) void wimage_refresh_kernel(const WImageDesc* __restrict__ descs, int n_desc) {
  __shared__ unsigned short tile[NS][32][36];     // 72-byte rows: 8-byte aligned quads, 18-dword stride spreads banks
  // binary search of the descriptor that owns this workgroup
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WImageDesc d = descs[lo];
  const int t = blockIdx.x - d.tile0;
  const int n0 = (t / d.tiles_k) * 32, k0 = (t % d.tiles_k) * 32;
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;     // 32 rows x 8 column quads
  const int n = n0 + r, k = k0 + c4;
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < d.N) {
    const float* src = d.w + (int64_t)n * d.ld + k;
    if (k + 4 <= d.K && (d.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
      x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < d.K) x[e] = src[e];
    }
  }
  uint32_t w0[NS], w1[NS];
  bf16_split2<NS>(x[0], x[1], w0);
  bf16_split2<NS>(x[2], x[3], w1);
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    // img[k / 32][n][k % 32]: the 64 bytes of row n in k-tile k0 / 32; columns k >= K receive zeros (x = 0 there)
    if (d.img && n < d.N)
      *reinterpret_cast<u32x2*>(d.img + p * d.plane + ((int64_t)(k0 >> 5) * d.N + n) * 32 + c4) = u32x2{w0[p], w1[p]};
    *reinterpret_cast<u32x2*>(&tile[p][r][c4]) = u32x2{w0[p], w1[p]};
  }
  if (d.imgT) {
    __syncthreads();
    const int kk = k0 + r;    // imgT[n / 32][k][n % 32]: thread (r, c4) writes row k0 + r, columns n0 + c4 .. + 3 (zeros for n >= N)
    if (kk < d.K) {
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const uint32_t lo = (uint32_t)tile[p][c4][r] | ((uint32_t)tile[p][c4 + 1][r] << 16);
        const uint32_t hi = (uint32_t)tile[p][c4 + 2][r] | ((uint32_t)tile[p][c4 + 3][r] << 16);
        if (d.imgT) *reinterpret_cast<u32x2*>(d.imgT + p * d.planeT + ((int64_t)(n0 >> 5) * d.K + kk) * 32 + c4) = u32x2{lo, hi};
      }
    }
  }
}
