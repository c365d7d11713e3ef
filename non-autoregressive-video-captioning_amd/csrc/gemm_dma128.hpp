// The DMA-fed member of the bf16 matrix-core GEMM family for the EXACT mode (three bf16 terms per fp32 operand, six matrix
// instructions per product block): nn.Linear forward and dX of models/bert.py:139-247, models/Encoder.py:9-66,
// models/__init__.py:83 and the NA pass decoding/algorithms.py:143-167 -- the launches that round 2's register-staged
// gemm_bf16_kernel<128 | 64> carried until round 5.
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]      Qop: fp32, k-contiguous (activations / dZ), live-row list;  Pop: pre-split image
//
// What the round-3 wide kernel (gemm_bf16_wide.hpp) showed to work, rebuilt for TWO workgroups per CU:
//   * both operands go global -> LDS by DMA (buffer_load_dwordx4 ... lds): no register staging, no LDS stores.  gemm_bf16_kernel
//     stores three bf16 planes of every activation tile (48 KB per 128 x 128 k-tile) and reads them back as fragments (96 KB) under
//     two barriers per k-tile; here Qop lands ONCE as raw fp32 (16 KB), Pop as a copy of the k-tile-major planes, and a k-tile
//     takes one barrier.  The live-row gather and the zero fill past the live rows / the reduce extent sit in the request
//     (per-lane offsets are k-invariant; the k-tile advances through the request's scalar offset).
//   * the exact three-way split of Qop happens on the FRAGMENTS, in registers (44 full-rate vector instructions per
//     32 x 16 fragment), only for the activation side: the weights were split once per step by nacf_wimage_refresh.
//   * v_mfma_f32_32x32x16_bf16, wave tile (32 MT) x 64: 6 MT + 6 fragment reads (ds_read_b128) and 44 MT split instructions per
//     12 MT matrix instructions of 32 cycles -- a third of the issue slots, where the 16 x 16 x 32 body of gemm_g256w.hpp needs
//     all of them.
//   * one workgroup is 4 waves (one per SIMD), 64 / 80 KB of LDS: TWO workgroups per CU, so the second one's k-loop covers the
//     first one's prologue (DMA latency), epilogue and barrier waits.  The k-loop is a software pipeline over k-steps pinned slot
//     by slot (see "the k-loop" below): the first form left the order to hipcc and to the partner wave on the SIMD, and neither
//     the split block nor the burst of DMA requests overlapped with the partner's matrix instructions.
//   * accumulators per (row block, column block) add their (k-tile, k-step, term) products in the wide kernel's order with the
//     wide kernel's instruction: the results are the wide kernel's, bit for bit (tests/test_dma128_gpu.py), and the epilogues
//     are the family's (EpiLinear / EpiStore / EpiArgmax), dropout masks keyed by the element index.
//
// Tiles: MT = 2 -> 128 x 128 (LDS 2 x (16 + 24) KB), MT = 1 -> 64 x 128 (2 x (8 + 24) KB) for launches under one round of 128-row
// tiles.  Reduce extents need not be multiples of 32: the image is zero-padded to whole k-tiles, Qop chunks past the extent are
// requested out of range (zeros), and a chunk that straddles the extent is cleaned in LDS before the last k-tile is read.
#pragma once
#include "gemm_bf16.hpp"

namespace dma128 {

constexpr int BN = 128, BK = 32, NT = 2, WTN = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MT> struct Geo {
  static constexpr int BM = 64 * MT, WTM = 32 * MT;
  static constexpr int Q_IMG = BM * 128;                    // [BM rows][32 fp32]
  static constexpr int P_PLANE = BN * 64;                   // [128 rows][32 bf16]
  static constexpr int P_IMG = 3 * P_PLANE;
  static constexpr int STAGE = Q_IMG + P_IMG;               // 40 / 32 KB
  static constexpr int LDS_BYTES = 2 * STAGE;               // 80 / 64 KB: two workgroups per CU
  static constexpr int QREQ = 2 * MT;                       // DMA requests (1 KB = 8 rows) per wave for one Qop tile
  static constexpr int PREQ = 6;                            // ... for the three planes of one Pop tile (24 requests of 16 rows)
};

// DMA128_GLOBAL (tuning builds; default 0): 1 = the requests are global_load_lds_dwordx4 (the wide kernel's instruction) instead of
// buffer_load_dwordx4 ... lds with per-lane offsets.  Tried because the counters of this kernel on the vocabulary shape
// (tools/pmc_gemm_cache2.sh, profiles/r06_dma128_cache_counters.txt) show TCP_TOTAL_CACHE_ACCESSES 67 M for 956 MB (one look-up per
// 16-byte lane element) and TA_BUSY 3 x the register-staged kernel's (21 M) and the wide kernel's (18 M for 774 MB): the same k-tile
// time, bit-identical results (3568 vs 3588 cycles, profiles/r06_dma128_probe_global_vs_buffer.txt) -- the instruction form is not
// what makes the difference.  The global form has no out-of-range zero fill: rows past the live count repeat the last live row
// (their outputs are never stored), chunks past the reduce extent are fetched from the k-tile's first chunk and the whole tail is
// zeroed in LDS.
#ifndef DMA128_GLOBAL
#define DMA128_GLOBAL 0
#endif
__device__ __forceinline__ void dma16g(uint32_t voff, uint32_t m0, const void* base) {      // LDS destination m0 + lane * 16
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(m0), "s"(base) : "memory");
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// 16-byte chunk swizzle of the fp32 image (128-byte rows, 8 chunks): the two ds_read_b128 of a 32-row fragment (lane = (row & 31,
// h): chunks 4s + 2h and + 1) are conflict-free under the instruction's lane groups (the wide kernel's fq)
__device__ __forceinline__ int fq(int row) { return (row >> 1) & 7; }

// per (column tile, row): max logit, its index (smallest on ties), sum exp(l - max) -- argmax_epilogue (gemm_f32.hpp) on the
// 32 x 32 accumulator map: register r of acc[a][b] is row a*32 + l31, column b*32 + 8*(r >> 2) + 4*lh + (r & 3) of the wave tile
template <int MT>
__device__ __forceinline__ void argmax_epilogue(float* smem, const GemmShape& g, const EpiArgmax& epi, f32x16 (&acc)[MT][NT], int m0, int n0,
                                                int Meff, int tile_n, int wm, int wn, int l31, int lh, int tid) {
  constexpr int BM = Geo<MT>::BM, WTM = Geo<MT>::WTM, WN = 2;
  float* redv = smem;                  // [WN][BM]
  float* reds = smem + WN * BM;        // [WN][BM]
  int* redi = reinterpret_cast<int*>(smem + 2 * WN * BM);
  const float NEG = -3.0e38f;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    float best = NEG;
    int bidx = 0x7fffffff;
    const int row = wm * WTM + a * 32 + l31;
    const int mrow = m0 + row;
    float* crow = nullptr;
    if (epi.C && mrow < Meff) crow = epi.C + (int64_t)(g.rows ? g.rows[mrow] : mrow) * epi.ldc;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = n0 + wn * WTN + b * 32 + 8 * j + 4 * lh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = nb + e;
          float v = NEG;
          if (n < g.N) v = acc[a][b][4 * j + e] + (epi.bias ? epi.bias[n] : 0.f);
          acc[a][b][4 * j + e] = v;
          if (v > best) { best = v; bidx = n; }
        }
        if (crow) {
          if (nb + 3 < g.N) *reinterpret_cast<f32x4*>(crow + nb) = f32x4{acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]};
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < g.N) crow[nb + e] = acc[a][b][4 * j + e];
          }
        }
      }
    }
    {
      const float ov = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bidx, 32, 64);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lh == 0) { redv[wn * BM + row] = best; redi[wn * BM + row] = bidx; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int row = wm * WTM + a * 32 + l31;
    float tmax = redv[row];
#pragma unroll
    for (int w = 1; w < WN; ++w) tmax = fmaxf(tmax, redv[w * BM + row]);
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[a][b][e];
        s += (v > -1.0e38f) ? (epi.C ? expf(v - tmax) : __expf(v - tmax)) : 0.f;
      }
    s += __shfl_xor(s, 32, 64);
    if (lh == 0) reds[wn * BM + row] = s;
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
      const float ov = redv[w * BM + row];
      const int oi = redi[w * BM + row];
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
      s += reds[w * BM + row];
    }
    const int64_t o = (int64_t)tile_n * g.M + m;
    epi.pmax[o] = best;
    epi.psum[o] = s;
    epi.pidx[o] = bidx;
  }
}

#ifdef DMA128_TRACE
__device__ unsigned long long* g_trace = nullptr;      // tuning builds (tools/probes/dma128_probe.hip): [workgroup][4] shader-clock stamps of wave 0
#define DMA128_MARK(i) do { if (g_trace && tid == 0) g_trace[(size_t)blockIdx.x * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define DMA128_MARK(i) do { } while (0)
#endif

// ABL: tuning builds only: 1 = no DMA inside the k-loop, 2 = no operand split, 4 = no matrix instructions, 32 = every tile requested twice, 64 / 128 = nt policy on the Qop / Pop requests
template <int MT, class Epi, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_dma128_kernel(GemmShape g, Epi epi) {
  using G = Geo<MT>;
  constexpr int BM = G::BM, WTM = G::WTM, QREQ = G::QREQ, PREQ = G::PREQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const int bid = blockIdx.x, z = blockIdx.z;
  DMA128_MARK(0);

  // ---- live rows, tile of this workgroup (as gemm_bf16_body)
  int Meff = g.M;
  if (g.count) Meff = min(Meff, *g.count);
  const int tiles_m_live = (Meff + BM - 1) / BM;
  const int nwg = tiles_m_live * g.tiles_n;
  if (bid >= nwg) {
    if (g.zero_dead && g.rows && z == 0) {      // workgroups past the live tiles zero-fill the dead rows of the output
      const int dt = bid - nwg;
      const int j0 = (dt / g.tiles_n) * BM, c0 = (dt % g.tiles_n) * BN;
      const int n_dead = g.M - Meff;
      for (int q = tid; q < BM * (BN / 4); q += 256) {
        const int j = j0 + q / (BN / 4), n = c0 + (q % (BN / 4)) * 4;
        if (j < n_dead && n < g.N) epi.zero4(g.rows[Meff + j], n, g.N);
      }
    }
    return;
  }
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
  int tile_m = logical / g.tiles_n, tile_n = logical % g.tiles_n;
  if (g.group_n > 0) {
    const int per = tiles_m_live * g.group_n;
    const int grp = logical / per, r = logical - grp * per;
    const int gn = min(g.group_n, g.tiles_n - grp * g.group_n);
    tile_m = r / gn;
    tile_n = grp * g.group_n + (r - tile_m * gn);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = z * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);      // (k_per_split is a multiple of 32: the launcher checks)
  const int klen = kend > kbeg ? kend - kbeg : 0;
  const int nk = __builtin_amdgcn_readfirstlane((klen + 31) >> 5);
  const int ktail = klen & 31;                  // != 0: the last k-tile holds only that many reduce indices

  // ---- descriptors and the k-invariant per-lane offsets
  const uint32_t a_bytes = (uint32_t)min((uint64_t)g.M * (uint64_t)g.ldq * 4u, (uint64_t)0x7ffffff0u);
  const uint32_t kt_bytes = (uint32_t)(g.ldpi * 2), plane_bytes = (uint32_t)(g.pimg_plane * 2);
  const uint32_t b_bytes = (uint32_t)((g.K + 31) >> 5) * kt_bytes + 2u * plane_bytes;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.Q, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.Pimg, 0, b_bytes, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;      // past the extent: the request returns zeros
  // Qop: request j of this wave = rows (wave QREQ + j) 8 .. + 7 of the tile; LDS chunk position lane & 7 of row R receives source
  // chunk (lane & 7) ^ fq(R).  Rows past the live count are requested out of range.
  uint32_t voa[4];
#pragma unroll
  for (int j = 0; j < QREQ; ++j) {
    const int R = (wave * QREQ + j) * 8 + (lane >> 3), m = m0 + R;
    const int mc = DMA128_GLOBAL ? min(m, Meff - 1) : m;
    const int ph = mc < Meff ? (g.rows ? g.rows[mc] : mc) : -1;
    voa[j] = ph >= 0 ? (uint32_t)ph * (uint32_t)(g.ldq * 4) + (uint32_t)(((lane & 7) ^ fq(R)) << 4) : OOB;
  }
  // the same for a last k-tile that holds only ktail reduce indices: chunks that begin at or past the extent arrive as zeros
  uint32_t voat[4];
  const int last_partial = ktail != 0 ? nk - 1 : -1;
  const bool tail_fix = DMA128_GLOBAL ? ktail != 0 : (ktail & 3) != 0;      // the last tile's image needs cleaning in LDS
#pragma unroll
  for (int j = 0; j < QREQ; ++j) {
    const int R = (wave * QREQ + j) * 8 + (lane >> 3);
    voat[j] = ((((lane & 7) ^ fq(R)) << 2) >= ktail) ? (DMA128_GLOBAL ? (voa[j] & ~0x70u) : OOB) : voa[j];
  }
  // Pop: request i of this wave = number r = wave 6 + i of the tile's 24: plane r >> 3, rows (r & 7) 16 .. + 15; rows past N repeat
  // row N - 1 (their columns are never stored)
  uint32_t vob[6];
#pragma unroll
  for (int i = 0; i < PREQ; ++i) {
    const int r = wave * PREQ + i, Rp = (r & 7) * 16 + (lane >> 2);
    vob[i] = (uint32_t)min(n0 + Rp, g.N - 1) * 64u + (uint32_t)(((lane & 3) ^ lds_sw(Rp)) << 4);
  }
  // the global form's sources (wave-uniform, in scalar registers) and the LDS base address
  const char* const q_src = reinterpret_cast<const char*>(uniform64(reinterpret_cast<uint64_t>(g.Q + kbeg)));
  const char* const p_src = reinterpret_cast<const char*>(uniform64(reinterpret_cast<uint64_t>(g.Pimg + (int64_t)(kbeg >> 5) * g.ldpi)));
  const int64_t p_tile_bytes = (int64_t)uniform64((uint64_t)(g.ldpi * 2)), p_plane_bytes = (int64_t)uniform64((uint64_t)(g.pimg_plane * 2));
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // ---- fragment addresses (byte offsets inside a stage)
  //   Pop (a operand, rows n): row wn 64 + b 32 + l31, 16-byte chunk 2s + lh of the plane's 64-byte row
  //   Qop (b operand, rows m): row wm WTM + a 32 + l31, fp32 chunks 4s + 2lh and + 1 of the 128-byte row
  const uint32_t p_rd0 = (uint32_t)(G::Q_IMG + (wn * WTN + l31) * 64 + ((lh ^ lds_sw(l31)) << 4));      // s = 0; s = 1: ^ 32
  const uint32_t q_row = (uint32_t)((wm * WTM + l31) * 128);
  const uint32_t qch = (uint32_t)((2 * lh) ^ fq(l31));

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- the k-loop: a software pipeline over k-STEPS (16 reduce indices = 12 MT matrix instructions per wave), pinned slot by
  //      slot.  Measured on the first, compiler-ordered form of this kernel (tools/probes/dma128_probe.hip, 2 workgroups per CU,
  //      cycles per k-tile and workgroup): matrix instructions alone 3180 (= 2 x 1536 + 3 %), + split 3790, + DMA 4310 -- neither
  //      the split block nor the burst of 10 DMA requests per wave behind the barrier (the four waves queue up at the address
  //      unit) overlapped with the partner wave's matrix instructions.  So every other instruction rides in the issue slots
  //      BEHIND one of this wave's own matrix instructions (32 cycles of pipe time, 4 of issue), as in the wide kernel:
  //        k-step X runs its 12 MT matrix instructions on fragment set X & 1 while it
  //          - reads the fragments of k-step X + 1 (2 MT raw fp32 + 6 plane reads, one ds_read_b128 per slot),
  //          - splits the raw fragments into set (X + 1) & 1 (44 MT vector instructions, 4 per slot),
  //          - (odd k-steps) requests tile t + 2 into the image tile t has just left (one request per slot).
  //      Tile t's image is read during k-steps (t - 1, 1) and (t, 0); the barrier between (t, 0) and (t, 1) frees it and
  //      publishes tile t + 1, requested one k-tile earlier.
  constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};     // six cross terms, smallest first
  constexpr int NSLOT = 12 * MT;
  u32x4 pf[2][NT][3];                 // Pop fragments [set][column block][plane]
  u32x4 qf[2][MT][3];                 // split Qop fragments [set][row block][plane]
  f32x4 raw[MT][2];                   // raw fp32 Qop fragment of the NEXT k-step [row block][half]
  float tmp[4];
  // vector operation `op` (0..43) of the exact three-way split (bf16_split2<3>) of raw[a] into q[a][.]; the four pairs advance
  // together so that neighbouring instructions are independent:
  //   0-3 pack h | 4-19 first residual (and x4, sub x4 for the even elements, then the odd ones) | 20-23 pack m |
  //   24-39 second residual | 40-43 pack l
  // (every result goes through an empty volatile asm: without it LLVM sinks the split of a whole k-step to where the next k-step
  //  uses it -- the scheduling fences below only bind the machine scheduler -- and the vector block runs with no matrix
  //  instruction to hide behind)
  auto split_op = [&](u32x4 (&q)[MT][3], const int a, const int op) __attribute__((always_inline)) {
    auto pack = [&](int plane, int pr) {
      uint32_t w = bf16_pack_top(raw[a][pr >> 1][(pr & 1) * 2], raw[a][pr >> 1][(pr & 1) * 2 + 1]);
      asm volatile("" : "+v"(w));
      q[a][plane][pr] = w;
    };
    if (op < 4) pack(0, op);
    else if (op < 20 || (op >= 24 && op < 40)) {
      const int o = (op < 20) ? op - 4 : op - 24;       // [element 0 | 1][and | sub][pair]
      const int el = o >> 3, sub = (o >> 2) & 1, pr = o & 3;
      const float x = raw[a][pr >> 1][(pr & 1) * 2 + el];
      if (!sub) { float w = f32_top16(x); asm volatile("" : "+v"(w)); tmp[pr] = w; }
      else { float w = x - tmp[pr]; asm volatile("" : "+v"(w)); raw[a][pr >> 1][(pr & 1) * 2 + el] = w; }
    }
    else if (op < 24) pack(1, op - 20);
    else pack(2, op - 40);
  };
  // fragment read number `i` of k-step s (0 | 1) of the image at `base` into set `SET`: i < 2 MT: raw half (i & 1) of row block i >> 1;
  // then plane reads (column block, plane)
  auto frag_read = [&](auto set_c, const unsigned char* base, const int s, const int i) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    if (i < 2 * MT) {
      const int a = i >> 1, h = i & 1;
      raw[a][h] = *reinterpret_cast<const f32x4*>(base + q_row + a * 4096 + ((qch ^ (uint32_t)(4 * s + h)) << 4));
    } else {
      const int j = i - 2 * MT, b = j / 3, pl = j % 3;
      pf[SET][b][pl] = *reinterpret_cast<const u32x4*>(base + (p_rd0 ^ (uint32_t)(32 * s)) + pl * G::P_PLANE + b * 2048);
    }
  };
  // one DMA request (number r of this wave's QREQ + PREQ) of k-tile kt into image `buf`
  auto issue_one = [&](const int kt, const int buf, const int r) __attribute__((always_inline)) {
    unsigned char* base = smem + buf * G::STAGE;
    if (r < QREQ) {
      const uint32_t vo = (kt == last_partial) ? voat[r] : voa[r];      // (a select, not a branch: the k-step stays one basic block)
      if constexpr (DMA128_GLOBAL && !(ABL & 192))
        dma16g(vo, lds0 + buf * G::STAGE + (wave * QREQ + r) * 1024, q_src + (int64_t)kt * 128);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + (wave * QREQ + r) * 1024), 16, vo, (uint32_t)(kbeg + 32 * kt) * 4u, 0, (ABL & 64) ? 2 : 0);
    } else {
      const int q = wave * PREQ + (r - QREQ);
      if constexpr (DMA128_GLOBAL && !(ABL & 192))
        dma16g(vob[r - QREQ], lds0 + buf * G::STAGE + G::Q_IMG + (q >> 3) * G::P_PLANE + (q & 7) * 1024,
               p_src + (int64_t)kt * p_tile_bytes + (int64_t)(q >> 3) * p_plane_bytes);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(base + G::Q_IMG + (q >> 3) * G::P_PLANE + (q & 7) * 1024), 16, vob[r - QREQ],
                                                 (uint32_t)((kbeg >> 5) + kt) * kt_bytes + (uint32_t)(q >> 3) * plane_bytes, 0, (ABL & 128) ? 2 : 0);
    }
  };
  // k-step S of the k-tile in image BUF: matrix instructions on fragment set S, everything of the next k-step behind them.
  // DMA (compile time): also request k-tile dma_kt (= this tile + 2) into image BUF, which this tile has just left
  auto kstep = [&](auto buf_c, auto s_c, auto dma_c, const int dma_kt) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value, S = decltype(s_c)::value;
    constexpr bool DMA = decltype(dma_c)::value;
    constexpr int NXT = S ^ 1;                                          // fragment set of the next k-step
    constexpr int NBUF = S ? (BUF ^ 1) : BUF;                           // its image: (t, 1) lives in tile t's, (t + 1, 0) in the other
    const unsigned char* nbase = smem + NBUF * G::STAGE;
    constexpr int NREAD = 2 * MT + 6;
    constexpr int SPLIT0 = MT == 2 ? 2 : 1;                             // first slot with split operations (4 per slot)
#pragma unroll
    for (int gs = 0; gs < NSLOT; ++gs) {
      const int b = gs / (6 * MT), t = (gs / MT) % 6, a = gs % MT;
      if constexpr (ABL & 4) asm volatile("" : "+v"(acc[a][b]) : "v"(pf[S][b][TP[t]]), "v"(qf[S][a][TQ[t]]));
      else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pf[S][b][TP[t]]), __builtin_bit_cast(bf16x8_t, qf[S][a][TQ[t]]), acc[a][b], 0, 0, 0);
      // fragment reads of the next k-step: the raw halves first (their split starts two slots later), then the planes
      if (MT == 2) { if (gs < NREAD) frag_read(std::integral_constant<int, NXT>{}, nbase, NXT, gs); }
      else { if (gs == 0) { frag_read(std::integral_constant<int, NXT>{}, nbase, NXT, 0); frag_read(std::integral_constant<int, NXT>{}, nbase, NXT, 1); }
             else if (gs < NREAD - 1) frag_read(std::integral_constant<int, NXT>{}, nbase, NXT, gs + 1); }
      if constexpr (!(ABL & 2)) {
        if (gs >= SPLIT0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int o = (gs - SPLIT0) * 4 + i;
            if (o < 44 * MT) split_op(qf[NXT], o / 44, o % 44);
          }
        }
      } else {
        if (gs == NSLOT - 1) {
#pragma unroll
          for (int a2 = 0; a2 < MT; ++a2) { qf[NXT][a2][0] = __builtin_bit_cast(u32x4, raw[a2][0]); qf[NXT][a2][1] = __builtin_bit_cast(u32x4, raw[a2][1]); qf[NXT][a2][2] = qf[NXT][a2][0] ^ qf[NXT][a2][1]; }
        }
      }
      if constexpr (DMA) {
        // requests of tile dma_kt, one per slot from the first slot behind the barrier on: they have the rest of this k-step and the
        // next one to land (one per second slot: the last ones arrived after the barrier that needs them, +250 cycles per k-tile)
        if (gs < QREQ + PREQ) issue_one(dma_kt, BUF, gs);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto publish = [&]() __attribute__((always_inline)) {      // this wave's requests have landed; every wave's: behind the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto clean_tail = [&](const int buf) __attribute__((always_inline)) {
    // the chunk that straddles the reduce extent carries 1-3 elements of whatever follows in the row: zero them in place
    if (tid < BM) {
      // (global form: every element from the extent on -- the chunks past it hold a copy of the k-tile's first chunk)
      const int kend_fix = DMA128_GLOBAL ? 32 : ((ktail >> 2) << 2) + 4;
      for (int k = ktail; k < kend_fix; ++k)
        reinterpret_cast<float*>(smem + buf * G::STAGE + tid * 128 + (((k >> 2) ^ fq(tid)) << 4))[k & 3] = 0.f;
    }
    __syncthreads();
  };
  // (DMA is a compile-time flag and the loop below has no run-time choice between k-step variants: where two variants merged,
  //  hipcc gave the 64 accumulators different registers on the two paths and copied them once per k-tile)
  auto ktile = [&](auto buf_c, auto dma_c, const int kt) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    // (ABL & 32, tuning: the same tile requested a second time from k-step 0 -- twice the bytes in flight; results undefined)
    kstep(buf_c, std::integral_constant<int, 0>{}, std::integral_constant<bool, (ABL & 32) != 0>{}, kt < nk - 2 ? kt + 2 : kt);
    publish();                                                   // tile kt + 1 is in image BUF ^ 1; image BUF is free
    if (tail_fix && kt + 2 == nk) clean_tail(BUF ^ 1);
    kstep(buf_c, std::integral_constant<int, 1>{}, dma_c, kt + 2);
  };

  // ---- prologue: tiles 0 and 1 requested together; the fragments of k-step (0, 0) are read and split with nothing to hide behind
  if (nk > 0) {
#pragma unroll
    for (int r = 0; r < QREQ + PREQ; ++r) issue_one(0, 0, r);
    if (nk > 1) {
#pragma unroll
      for (int r = 0; r < QREQ + PREQ; ++r) issue_one(1, 1, r);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QREQ + PREQ) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tail_fix && nk == 1) clean_tail(0);
#pragma unroll
    for (int i = 0; i < 2 * MT + 6; ++i) frag_read(std::integral_constant<int, 0>{}, smem, 0, i);
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int o = 0; o < 44 * MT; ++o) split_op(qf[0], o / 44, o % 44);
    } else {
#pragma unroll
      for (int a2 = 0; a2 < MT; ++a2) { qf[0][a2][0] = __builtin_bit_cast(u32x4, raw[a2][0]); qf[0][a2][1] = __builtin_bit_cast(u32x4, raw[a2][1]); qf[0][a2][2] = qf[0][a2][0] ^ qf[0][a2][1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    DMA128_MARK(1);
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    constexpr bool DMA_ON = !(ABL & 1);
    using DY = std::integral_constant<bool, DMA_ON>;
    using DN = std::false_type;
    int kt = 0;
    // steady pairs: both tiles have a tile two ahead to request
#pragma nounroll
    for (; kt + 3 < nk; kt += 2) {
      ktile(B0{}, DY{}, kt);
      ktile(B1{}, DY{}, kt + 1);
    }
    // the last one to three tiles, straight line
    if (kt + 2 < nk) { ktile(B0{}, DY{}, kt); ktile(B1{}, DN{}, kt + 1); ktile(B0{}, DN{}, kt + 2); }
    else if (kt + 1 < nk) { ktile(B0{}, DN{}, kt); ktile(B1{}, DN{}, kt + 1); }
    else ktile(B0{}, DN{}, kt);
  }

  DMA128_MARK(2);
  // ---- epilogue: register r of acc[a][b] is row (m) a*32 + l31, column (n) b*32 + 8*(r >> 2) + 4*lh + (r & 3) of the wave tile
  //      (the matrix instruction is issued with a = Pop fragment, b = Qop fragment): 4 float4s per 32 x 32 block and row
  if constexpr (!Epi::kArgmax) {
    constexpr int TN = NT * 4;
    int ncol[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) ncol[b] = n0 + wn * WTN + (b >> 2) * 32 + (b & 3) * 8 + 4 * lh;
    const bool fast = epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N;
    // one row block at a time, the block index a COMPILE-TIME constant: inside a run-time loop (hipcc declines to unroll around the
    // inlined epilogues) the accumulator array is indexed dynamically and lives in scratch for the whole kernel
    auto rowblock = [&](auto a_c) __attribute__((always_inline)) {
      constexpr int a = decltype(a_c)::value;
      f32x4 acc4[1][TN];
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc4[0][b] = f32x4{acc[a][b >> 2][4 * (b & 3)], acc[a][b >> 2][4 * (b & 3) + 1], acc[a][b >> 2][4 * (b & 3) + 2], acc[a][b >> 2][4 * (b & 3) + 3]};
      int mlog[1] = {m0 + wm * WTM + a * 32 + l31};
      int mphys[1] = {(g.rows && mlog[0] < Meff) ? g.rows[mlog[0]] : mlog[0]};
      if (fast) epi.template tile_fast<1, TN, true>(acc4, mphys, ncol, g.N, z);
      else epilogue_all<0, 1, TN, true, Epi>(epi, acc4, mlog, mphys, ncol, Meff, g.N, z);
    };
    rowblock(std::integral_constant<int, 0>{});
    if constexpr (MT == 2) rowblock(std::integral_constant<int, 1>{});
  } else {
    __syncthreads();      // the last tile's fragments are read: the scratch may reuse the images
    argmax_epilogue<MT>(reinterpret_cast<float*>(smem), g, epi, acc, m0, n0, Meff, tile_n, wm, wn, l31, lh, tid);
  }
#ifdef DMA128_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DMA128_MARK(3);
#endif
}

// host side: eligibility and launch
inline bool eligible(const GemmShape& g, int splits) {
  if (!g.Pimg) return false;
  if (splits > 1 && (g.k_per_split & 31) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(g.Q) & 15) != 0 || (g.ldq & 3) != 0) return false;
  if ((uint64_t)g.M * (uint64_t)g.ldq * 4u >= 0x7ffffff0ull) return false;                          // 31-bit request offsets
  const uint64_t bb = (uint64_t)((g.K + 31) >> 5) * (uint64_t)g.ldpi * 2u + 2u * (uint64_t)g.pimg_plane * 2u;
  return bb < 0x7ffffff0ull && (uint64_t)g.N * 64u < 0x7ffffff0ull;
}
template <int MT, class Epi>
inline void launch(GemmShape g, const Epi& epi, int splits, hipStream_t s) {
  auto kern = gemm_dma128_kernel<MT, Epi>;
  constexpr int lds = Geo<MT>::LDS_BYTES;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  g.tiles_m = cdiv(g.M, Geo<MT>::BM);
  g.tiles_n = cdiv(g.N, BN);
  g.group_n = 0;
  {
    // L2-aware order for very wide P (GemmShape::group_n), as the 128 x 128 kernel's launcher
    static const int group_n = [] { const char* e = getenv("NACF_GEMM_GROUP_N"); return e ? atoi(e) : 6; }();
    if (group_n > 0 && g.tiles_n >= 32 && splits == 1) g.group_n = group_n;
  }
  dim3 grid((g.tiles_m + (g.zero_dead ? 1 : 0)) * g.tiles_n, 1, splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, epi);
}

}  // namespace dma128
