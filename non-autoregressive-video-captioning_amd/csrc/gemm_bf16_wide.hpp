// The "wide" member of the bf16 matrix-core GEMM family (gemm_bf16.hpp): same operand conventions, row sets, split-K and
// epilogues, same exact three-way split, but built for ONE workgroup per CU:
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]      Qop: fp32, k-contiguous (activations / dZ);  Pop: pre-split bf16 image
//
//   * 4 waves (one per SIMD), WAVE tile 64 (m) x 128 (n), workgroup tile 128 x 256, v_mfma_f32_32x32x16_bf16.
//     tools/probes/compute_phase.hip showed the 64 x 64 wave tile of gemm_bf16_kernel bound by its LDS fragment reads (24
//     ds_read_b128 per k-tile per 1536 matrix cycles); here a wave reads 32 per 3072 (Q is read as raw fp32).  A lone wave
//     issues about one instruction per 4-7 cycles, so with the 16-cycle 16x16x32 instruction the k-tile was ISSUE bound
//     (4700 cycles for 192 MFMAs + 176 split operations + 32 reads + 16 DMA requests; tools/probes/wide_gemm.hip); the
//     32-cycle shape halves the matrix instructions and leaves ~2.6 other instructions per 32-cycle slot.
//   * no register staging and no LDS stores: both operands go global -> LDS by DMA (global_load_lds_dwordx4), the XOR
//     swizzles applied to the per-lane SOURCE address (the LDS image of a DMA is lane-linear).  P is the k-tile-major
//     image (a pure copy); Q lands as raw fp32 and is split into its three bf16 terms on the FRAGMENTS, in registers, two
//     vector instructions per matrix instruction, one k-tile ahead of its use.
//   * one barrier per k-tile, placed before the 4th (last) column block: every wave holds that block's P fragments in
//     registers by then, so the barrier frees P's image for the DMA of tile t+2 while the 24 MFMAs of the last block cover
//     the first fragment reads of tile t+1.  DMA requests are spread one per 4 matrix instructions (issued back to back
//     the four waves queue up behind the address unit: 78 cycles per request).  Q has three images (its tiles are
//     requested three tiles ahead: activations come from HBM, weights from L2).
//   * matrix instructions, LDS reads, DMA requests and waits are inline assembly in program order: left to the builtins,
//     hipcc parks accumulators in a 16-register window (hundreds of v_accvgpr moves per k-tile) and cannot know which
//     LDS reads a DMA in flight may alias.
//   LDS: 3 x 16 KB (Q, fp32) + 2 x 48 KB (P, three planes) = 144 KB.
#pragma once
#include "gemm_bf16.hpp"

namespace wide {
constexpr int BN = 256, BK = 32;
constexpr int WM = 2, WN = 2, WTN = 128, NT = WTN / 32;      // per wave: MT x 4 accumulator tiles of 32 x 32
constexpr int P_PLANE = BN * 64;                             // [256 rows][32 bf16]
constexpr int P_STAGE = 3 * P_PLANE;
// MT = 32-row blocks per wave: 2 -> workgroup tile 128 x 256 (the fewest LDS / DMA bytes per matrix instruction),
//                              1 -> 64 x 256 (twice the workgroups: launches of 120 .. 250 big tiles)
template <int MT> struct Geo {
  static constexpr int BM = 64 * MT, WTM = 32 * MT;
  static constexpr int Q_STAGE = BM * 128;                   // [BM rows][32 fp32]
  static constexpr int Q_BASE = 0, P_BASE = 3 * Q_STAGE;
  static constexpr int LDS_BYTES = 3 * Q_STAGE + 2 * P_STAGE;
  static constexpr int QREQ = 2 * MT;                        // DMA requests per wave for one Q tile
  static constexpr int NSLOT = 48 * MT, PH = 12 * MT;        // matrix instructions per k-tile / per column block
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef WIDE_TRACE
__device__ int g_wide_skew = 0;      // tuning: start delay of workgroup b = ((b >> 3) & 7) * g_wide_skew * 64 cycles
__device__ unsigned long long* g_wide_trace = nullptr;     // tuning builds: [workgroup][4] shader-clock stamps of wave 0
#define WIDE_MARK(i) do { if (g_wide_trace && tid == 0) g_wide_trace[(size_t)bid * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WIDE_MARK(i) do { } while (0)
#endif

// 16-byte chunk swizzle of the fp32 Q image (128-byte rows, 8 chunks): the two ds_read_b128 of a 32-row fragment (lane =
// (row & 31, h): chunks 4s + 2h and + 1) are conflict-free under the instruction's lane groups (MI355X_MICROARCH.md, LDS)
__device__ __forceinline__ int fq(int row) { return (row >> 1) & 7; }

// ---- the instructions the compiler must not touch
// The 8 accumulators (row block mt, column block nt) live in FIXED registers a[16 (4 mt + nt) ...]: as C++ values ("+a"
// operands) hipcc moved them between code regions with v_accvgpr copies placed right behind the matrix instruction that
// wrote them -- it does not know the asm statement is one, so it keeps no distance (wrong last rows).  Named only inside
// the assembly text and as clobbers they are never copied; what remains possible is that the compiler parks a spilled
// vector register in one of them where it sees no clobber nearby -- tools/check_wide_hazards.py fails the build on
// either pattern (csrc/Makefile).  mfma_fixed<ACC>, acc_zero_all, acc_read<ACC>:
#include "gemm_bf16_wide_regs.inc"
template <int OFF, class V>
__device__ __forceinline__ void lds_read(V& r, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
}
// global -> LDS, 16 bytes per lane: LDS destination m0 + lane * 16, source base + voff
__device__ __forceinline__ void dma16(uint32_t voff, uint32_t m0, const void* base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(m0), "s"(base) : "memory");
}
// a wave-uniform 64-bit value the compiler may hold in vector registers (a problem table read through scratch in the grouped
// launch) back into scalar registers: the DMA's base is an "s" operand
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void for_each_ic(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }

// ---------------------------------------------------------------- fused soft-max statistics (EpiArgmax), 32-shape map
// register r of acc[mt][nt]: row mt*32 + l31, column nt*32 + 8*(r >> 2) + 4*lh + (r & 3) of the wave tile; same outputs
// as argmax_epilogue (gemm_f32.hpp): per (column tile, row) max logit, its index (smallest on ties), sum exp(l - max)
template <int MT>
__device__ __forceinline__ void argmax_epilogue32(float* smem, const GemmShape& g, const EpiArgmax& epi, f32x16 (&acc)[MT][NT],
                                                  int m0, int n0, int Meff, int tile_n, int wm, int wn, int l31, int lh, int tid) {
  constexpr int BM = Geo<MT>::BM, WTM = Geo<MT>::WTM;
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

// ABL: tuning builds only (tools/probes/wide_gemm.hip): 1 = no DMA inside the k-loop, 2 = no Q split
template <int MT, class Epi, int ABL = 0>
__device__ __forceinline__ void gemm_wide_body(const GemmShape& g, const Epi& epi, const int bid, const int z, const int nz) {
  using G = Geo<MT>;
  constexpr int BM = G::BM, WTM = G::WTM, Q_STAGE = G::Q_STAGE, Q_BASE = G::Q_BASE, P_BASE = G::P_BASE, QREQ = G::QREQ;
  constexpr int NSLOT = G::NSLOT, PH = G::PH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  int Meff = g.M;
  if (g.count) Meff = min(Meff, *g.count);
  const int tiles_m_live = (Meff + BM - 1) / BM;
  const int nwg = tiles_m_live * g.tiles_n;
  if (bid >= nwg) {
    if (g.zero_dead && g.rows && z == 0) {
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
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = bid & 7, slot_x = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot_x;
  int tile_m = logical / g.tiles_n, tile_n = logical % g.tiles_n;
  if (g.group_n > 0) {
    const int per = tiles_m_live * g.group_n;
    const int grp = logical / per, r = logical - grp * per;
    const int gn = min(g.group_n, g.tiles_n - grp * g.group_n);
    tile_m = r / gn;
    tile_n = grp * g.group_n + (r - tile_m * gn);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = __builtin_amdgcn_readfirstlane(kend > kbeg ? (kend - kbeg) / BK : 0);        // the launcher guarantees whole k-tiles: an even count >= 4

  // ---- DMA sources: byte offsets per lane (32 bits: the launcher checks the operands' extents) from a base that moves
  //      one k-tile per trip.  Q: request j of this wave covers rows (wave*QREQ + j)*8 .. +7, lane -> (row, chunk position);
  //      P: per plane, request j covers rows (wave*4 + j)*16 .. +15 of the k-tile's 16 KB
  uint32_t qoff[QREQ], poff[4];
#pragma unroll
  for (int j = 0; j < QREQ; ++j) {
    const int R = (wave * QREQ + j) * 8 + (lane >> 3);
    const int gc = min(m0 + R, Meff - 1);
    const int ph = g.rows ? g.rows[gc] : gc;
    qoff[j] = (uint32_t)((int64_t)ph * g.ldq * 4) + 16 * ((lane & 7) ^ fq(R));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int Rp = (wave * 4 + j) * 16 + (lane >> 2);
    const int gn = min(n0 + Rp, g.N - 1);
    poff[j] = (uint32_t)gn * 64 + 16 * ((lane & 3) ^ lds_sw(Rp));
  }
  const char* const q_src = reinterpret_cast<const char*>(uniform64(reinterpret_cast<uint64_t>(g.Q + kbeg)));      // + kt * 128
  const char* const p_src = reinterpret_cast<const char*>(uniform64(reinterpret_cast<uint64_t>(g.Pimg + (int64_t)(kbeg >> 5) * g.ldpi)));   // + kt * tile bytes + plane * plane bytes
  const int64_t p_tile_bytes = (int64_t)uniform64((uint64_t)(g.ldpi * 2)), p_plane_bytes = (int64_t)uniform64((uint64_t)(g.pimg_plane * 2));
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const uint32_t dma_q_m0 = lds0 + Q_BASE + wave * (QREQ * 1024);     // + image * Q_STAGE + j * 1024
  const uint32_t dma_p_m0 = lds0 + P_BASE + wave * 4096;              // + image * P_STAGE + plane * P_PLANE + j * 1024
  auto dma_q_one = [&](int img, int j, int kt) __attribute__((always_inline)) {
    dma16(qoff[j], dma_q_m0 + img * Q_STAGE + j * 1024, q_src + (int64_t)kt * (BK * 4));
  };
  auto dma_p_one = [&](int img, int i, int kt) __attribute__((always_inline)) {      // i = plane * 4 + j
    const int p = i >> 2, j = i & 3;
    dma16(poff[j], dma_p_m0 + img * P_STAGE + p * P_PLANE + j * 1024, p_src + kt * p_tile_bytes + p * p_plane_bytes);
  };

  // ---- fragment addresses (LDS byte addresses of this lane)
  //   P (a operand, rows n): row wn*128 + nt*32 + l31, 16-byte chunk 2s + lh of the plane's 64-byte row
  //   Q (b operand, rows m): row wm*WTM + mt*32 + l31, fp32 chunks 4s + 2lh and + 1 of the 128-byte row
  const uint32_t p_rd0 = lds0 + P_BASE + (wn * WTN + l31) * 64 + ((lh ^ lds_sw(l31)) * 16);      // s = 0; + image, plane, nt*2048
  const uint32_t p_rd1 = p_rd0 ^ 32;                                                              // s = 1
  const uint32_t p_rd[2][2] = {{p_rd0, p_rd1}, {p_rd0 + P_STAGE, p_rd1 + P_STAGE}};               // [image][s] (ds offsets are 16 bits)
  const uint32_t q_row = lds0 + Q_BASE + (wm * WTM + l31) * 128;
  const int qch = (2 * lh) ^ fq(l31);
  const uint32_t q_rd00 = q_row + qch * 16, q_rd01 = q_row + (qch ^ 1) * 16;                     // [s][r]; + image, mt*4096
  const uint32_t q_rd10 = q_row + (qch ^ 4) * 16, q_rd11 = q_row + (qch ^ 5) * 16;

  u32x4 qa[MT][2][3], qb[MT][2][3];   // split Q fragments [mt][s][plane]: one set in use, the other being built
  u32x4 pf[2][2][3];                  // P fragments of one column block [ping-pong][s][plane]
  f32x4 raw[2][2];                    // raw fp32 Q fragment being split: [unit parity][half]
  float tmp[4];

  // read number I = s * 3 + plane (of 6) of the P fragments of column block NTI from image IMG into pf[SET]
  auto read_pf_one = [&](auto set_c, auto img_c, auto nt_c, auto i_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, IMG = decltype(img_c)::value, NTI = decltype(nt_c)::value, I = decltype(i_c)::value;
    constexpr int S = I / 3, PL = I % 3;
    lds_read<PL * P_PLANE + NTI * 2048>(pf[SET][S][PL], p_rd[IMG][S]);
  };
  // read R of unit U = mt * 2 + s of the next tile's Q, from the image at byte offset qimg (run time: three images)
  auto read_raw = [&](auto u_c, auto r_c, uint32_t qimg) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value, R = decltype(r_c)::value;
    constexpr int S = U & 1;
    lds_read<(U >> 1) * 4096>(raw[U & 1][R], (S ? (R ? q_rd11 : q_rd10) : (R ? q_rd01 : q_rd00)) + qimg);
  };
  // vector operation `op` (0..43) of the exact three-way split (bf16_split2) of unit u into q[mt][s][.].  The four pairs
  // of a unit advance together so that neighbouring instructions are independent:
  //   0-3 pack h | 4-19 first residual (and x4, sub x4 for the even elements, then the odd ones) | 20-23 pack m |
  //   24-39 second residual | 40-43 pack l
  auto split_op = [&](u32x4 (&q)[MT][2][3], int u, int op) __attribute__((always_inline)) {
    const int mt = u >> 1, s = u & 1;
    auto pack = [&](int plane, int pr) {
      q[mt][s][plane][pr] = bf16_pack_top(raw[u & 1][pr >> 1][(pr & 1) * 2], raw[u & 1][pr >> 1][(pr & 1) * 2 + 1]);
    };
    if (op < 4) pack(0, op);
    else if (op < 20 || (op >= 24 && op < 40)) {
      const int o = (op < 20) ? op - 4 : op - 24;       // [element 0 | 1][and | sub][pair]
      const int el = o >> 3, sub = (o >> 2) & 1, pr = o & 3;
      const float x = raw[u & 1][pr >> 1][(pr & 1) * 2 + el];
      if (!sub) tmp[pr] = f32_top16(x);
      else raw[u & 1][pr >> 1][(pr & 1) * 2 + el] = x - tmp[pr];
    }
    else if (op < 24) pack(1, op - 20);
    else pack(2, op - 40);
  };
  constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};     // six cross terms, smallest first

  // ---- one k-tile: NSLOT matrix-instruction slots = 4 column blocks x 2 k-steps x 6 terms x MT row blocks; after the
  //      matrix instruction of slot GS the slot's share of everything else (MT = 2 | MT = 1):
  //   the 6 fragment reads of the next column block (block 0 of the NEXT tile, other image, during the last block)
  //                                     slots PH p + 0..5
  //   raw reads of the next tile's Q    units 0..3 at 2,3 | 26,27 | 44,45 | 66,67       units 0..1 at 2,3 | 14,15
  //   waits (all LDS reads so far)      0, 10, 24, 32, 48, 72 (= the barrier)           0, 6, 12, 24, 36 (= the barrier)
  //   the split operations              slots 10..95, 2 per slot (3 in the last four)   unit 0: 6..20, unit 1: 24..38, 3 per slot
  //   DMA requests of P(t+1), 6..11     2,6,..,22                                        1,3,..,11
  //   DMA requests of Q(t+3)            26,30,34,38                                      13,15
  //   DMA requests of P(t+2), 0..5      74,78,..,94 (after the barrier freed the image)  37,39,..,47
  //   the barrier waits for all DMA requests but the QREQ newest (Q(t+3): needed a tile later)
  //   MODE: 4 = inside the steady loop; 0..3 = a final tile with that many tiles after it (requests past the end are
  //   dropped at compile time).  There is deliberately NO variant with run-time tests: its branches made hipcc spill
  //   vector registers into a0.., i.e. into the accumulators it does not know about (tools/check_wide_hazards.py)
  auto tile = [&](auto par_c, auto mode_c, const int kt, const int q3) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value, MODE = decltype(mode_c)::value;
    constexpr bool STEADY = MODE >= 3;                    // every request exists: the barrier may leave the newest in flight
    constexpr bool dma_p1 = MODE >= 1, dma_q = MODE >= 3, dma_p2 = MODE >= 2;
    auto& QC = PAR ? qb : qa;
    auto& QN = PAR ? qa : qb;
    const int q3n = q3 == 2 ? 0 : q3 + 1;                 // image of Q(t+1)
    const uint32_t qimg = q3n * Q_STAGE;
    auto slot = [&](auto gs_c) __attribute__((always_inline)) {
      constexpr int gs = decltype(gs_c)::value;
      constexpr int p = gs / PH, sl = gs % PH, s = sl / (6 * MT), term = (sl / MT) % 6, mt = sl % MT;
      // waits (all LDS reads issued so far have returned); the raw registers ride along so that no use moves above
      constexpr int RW0 = MT == 2 ? 10 : 6;               // first wait for raw[0] (the second one falls on a block start)
      if constexpr (gs == 0 || (gs == PH && MT == 2) || (gs == PH && MT == 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (gs == RW0 || (MT == 2 && gs == 2 * PH)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]) :: "memory");
      if constexpr ((MT == 2 && gs == 32) || (MT == 1 && gs == 2 * PH)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
      if constexpr (gs == 3 * PH) {
        if constexpr (STEADY && MT == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
        else if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" : "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      mfma_fixed<mt * 4 + p>(pf[p & 1][s][TP[term]], QC[mt][s][TQ[term]]);
      // ---- fragment reads of the next column block
      if constexpr (sl < 6) {
        if constexpr (p < 3) read_pf_one(IC<(p + 1) & 1>{}, IC<PAR>{}, IC<p + 1>{}, IC<sl>{});
        else read_pf_one(IC<0>{}, IC<PAR ^ 1>{}, IC<0>{}, IC<sl>{});
      }
      // ---- the next tile's Q: raw reads ...
      if constexpr (MT == 2) {
        if constexpr (gs == 2 || gs == 3) read_raw(IC<0>{}, IC<gs - 2>{}, qimg);
        if constexpr (gs == 26 || gs == 27) read_raw(IC<1>{}, IC<gs - 26>{}, qimg);
        if constexpr (gs == 44 || gs == 45) read_raw(IC<2>{}, IC<gs - 44>{}, qimg);
        if constexpr (gs == 66 || gs == 67) read_raw(IC<3>{}, IC<gs - 66>{}, qimg);
      } else {
        if constexpr (gs == 2 || gs == 3) read_raw(IC<0>{}, IC<gs - 2>{}, qimg);
        if constexpr (gs == 14 || gs == 15) read_raw(IC<1>{}, IC<gs - 14>{}, qimg);
      }
      // ---- ... and its split
      if constexpr (!(ABL & 2)) {
        if constexpr (MT == 2) {
          if constexpr (gs >= 10) {
            constexpr int first = gs < 92 ? (gs - 10) * 2 : 164 + (gs - 92) * 3, cnt = gs < 92 ? 2 : 3;
            for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int o = first + decltype(i_c)::value; split_op(QN, o / 44, o % 44); },
                        std::make_integer_sequence<int, cnt>{});
          }
        } else {
          if constexpr ((gs >= 6 && gs < 21) || (gs >= 24 && gs < 39)) {
            constexpr int u = gs >= 24 ? 1 : 0, first = (gs - (u ? 24 : 6)) * 3;
            for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int o = first + decltype(i_c)::value; if constexpr (o < 44) split_op(QN, u, o); },
                        std::make_integer_sequence<int, 3>{});
          }
        }
      }
      // ---- DMA requests
      if constexpr (!(ABL & 1)) {
        if constexpr (MT == 2) {
          if constexpr (gs < 24 && (gs & 3) == 2 && dma_p1) dma_p_one(PAR ^ 1, 6 + gs / 4, kt + 1);
          if constexpr (gs >= 24 && gs < 40 && (gs & 3) == 2 && dma_q) dma_q_one(q3, (gs - 24) / 4, kt + 3);
          if constexpr (gs >= 72 && (gs & 3) == 2 && dma_p2) dma_p_one(PAR, (gs - 72) / 4, kt + 2);
        } else {
          if constexpr (gs < 12 && (gs & 1) == 1 && dma_p1) dma_p_one(PAR ^ 1, 6 + gs / 2, kt + 1);
          if constexpr ((gs == 13 || gs == 15) && dma_q) dma_q_one(q3, (gs - 13) / 2, kt + 3);
          if constexpr (gs >= 37 && (gs & 1) == 1 && dma_p2) dma_p_one(PAR, (gs - 37) / 2, kt + 2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for_each_ic(slot, std::make_integer_sequence<int, NSLOT>{});
  };

#ifdef WIDE_TRACE
  for (int i = 0, n = ((bid >> 3) & 7) * g_wide_skew; i < n; ++i) __builtin_amdgcn_s_sleep(1);
#endif
  WIDE_MARK(0);
  {
    // ---- prologue: P(0), Q(0), Q(1), Q(2) and the first half of P(1) requested together; Q(0) is split with nothing to
    //      hide behind
#pragma unroll
    for (int i = 0; i < 12; ++i) dma_p_one(0, i, 0);
#pragma unroll
    for (int j = 0; j < QREQ; ++j) dma_q_one(0, j, 0);
#pragma unroll
    for (int j = 0; j < QREQ; ++j) dma_q_one(1, j, 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma_p_one(1, i, 1);
#pragma unroll
    for (int j = 0; j < QREQ; ++j) dma_q_one(2, j, 2);
    acc_zero_all<4 * MT>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_raw(IC<0>{}, IC<0>{}, 0); read_raw(IC<0>{}, IC<1>{}, 0);
    read_raw(IC<1>{}, IC<0>{}, 0); read_raw(IC<1>{}, IC<1>{}, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
#pragma unroll
    for (int op = 0; op < 44; ++op) split_op(qa, 0, op);
#pragma unroll
    for (int op = 0; op < 44; ++op) split_op(qa, 1, op);
    if constexpr (MT == 2) {
      read_raw(IC<2>{}, IC<0>{}, 0); read_raw(IC<2>{}, IC<1>{}, 0);
      read_raw(IC<3>{}, IC<0>{}, 0); read_raw(IC<3>{}, IC<1>{}, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
#pragma unroll
      for (int op = 0; op < 44; ++op) split_op(qa, 2, op);
#pragma unroll
      for (int op = 0; op < 44; ++op) split_op(qa, 3, op);
    }
    for_each_ic([&](auto i_c) __attribute__((always_inline)) { read_pf_one(IC<0>{}, IC<0>{}, IC<0>{}, i_c); }, std::make_integer_sequence<int, 6>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // everybody is done with Q image 0 (tile 0 requests Q(3) into it)
    asm volatile("" ::: "memory");
    WIDE_MARK(1);
    // ---- steady trips: two tiles each (register sets and P images alternate), every request exists, no branches
    int kt = 0, q3 = 0;
    auto next3 = [&]() { q3 = q3 == 2 ? 0 : q3 + 1; };
#pragma nounroll
    for (; kt + 4 < nk; kt += 2) {
      tile(IC<0>{}, IC<4>{}, kt, q3);
      next3();
      tile(IC<1>{}, IC<4>{}, kt + 1, q3);
      next3();
    }
    // ---- the last four tiles, straight line: requests past the end are dropped at compile time; the split of the "next"
    //      tile and the first fragment reads of the "next" tile run on stale data (results unused)
    tile(IC<0>{}, IC<3>{}, kt, q3); next3();
    tile(IC<1>{}, IC<2>{}, kt + 1, q3); next3();
    tile(IC<0>{}, IC<1>{}, kt + 2, q3); next3();
    tile(IC<1>{}, IC<0>{}, kt + 3, q3);
  }
  WIDE_MARK(2);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // the last matrix instruction has written its rows
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the fragment reads that ran past the last tile have returned)
  __syncthreads();      // the images are dead: the soft-max epilogue reuses them as scratch

  if constexpr (!Epi::kArgmax) {
    // ---- accumulator map of the 32 x 32 x 16 instruction issued with a = P fragment, b = Q fragment: register r of
    //      acc[mt][nt] is row (m) mt*32 + l31, column (n) nt*32 + 8*(r >> 2) + 4*lh + (r & 3): four consecutive columns per
    //      register quad, handed to the epilogues of gemm_f32.hpp as 16 float4s per row block with their own row / column
    //      maps.  One row block at a time: with all 128 values (plus the residuals and biases of a fused nn.Linear
    //      epilogue) in flight hipcc spilled into accumulation registers that had not been read out yet.
    constexpr int TN = NT * 4;
    int ncol[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) ncol[b] = n0 + wn * WTN + (b >> 2) * 32 + (b & 3) * 8 + 4 * lh;
    const bool fast = epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N;
    for_each_ic([&](auto a_c) __attribute__((always_inline)) {
      constexpr int A = decltype(a_c)::value;
      f32x16 acc[NT];
      for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int I = decltype(i_c)::value; acc_read<A * 4 + I>(acc[I]); },
                  std::make_integer_sequence<int, NT>{});
      f32x4 acc4[1][TN];
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc4[0][b] = f32x4{acc[b >> 2][4 * (b & 3)], acc[b >> 2][4 * (b & 3) + 1], acc[b >> 2][4 * (b & 3) + 2], acc[b >> 2][4 * (b & 3) + 3]};
      int mlog[1] = {m0 + wm * WTM + A * 32 + l31};
      int mphys[1] = {(g.rows && mlog[0] < Meff) ? g.rows[mlog[0]] : mlog[0]};
      if (fast) epi.template tile_fast<1, TN, true>(acc4, mphys, ncol, g.N, z);
      else epilogue_all<0, 1, TN, true, Epi>(epi, acc4, mlog, mphys, ncol, Meff, g.N, z);
    }, std::make_integer_sequence<int, MT>{});
  } else {
    f32x16 acc[MT][NT];
    for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int I = decltype(i_c)::value; acc_read<I>(acc[I / 4][I % 4]); },
                std::make_integer_sequence<int, 4 * MT>{});
    argmax_epilogue32<MT>(reinterpret_cast<float*>(smem_raw), g, epi, acc, m0, n0, Meff, tile_n, wm, wn, l31, lh, tid);
  }
  WIDE_MARK(3);
}

template <int MT, class Epi, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm_wide_kernel(GemmShape g, Epi epi) {
  gemm_wide_body<MT, Epi, ABL>(g, epi, (int)blockIdx.x, (int)blockIdx.z, (int)gridDim.z);
}
}  // namespace wide
