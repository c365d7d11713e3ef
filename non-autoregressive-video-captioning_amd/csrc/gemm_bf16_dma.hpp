// The throughput mode's (NS = 1) forward / dX GEMM without register staging: same operand conventions, row sets, split-K and
// epilogues as gemm_bf16.hpp,
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]      Qop: fp32, k-contiguous (activations / dZ);  Pop: the one-plane bf16 image
//
// but both operands go global -> LDS by DMA (global_load_lds_dwordx4) and nothing is converted on the way: P is the
// k-tile-major image (a pure copy), Q lands as raw fp32 and is rounded to bf16 (RNE, v_cvt_pk_bf16_f32: the same values the
// register-staged kernels feed the matrix cores) on the FRAGMENTS, four conversions per matrix instruction operand.
// Why: with one matrix instruction per product block instead of six, the register-staged 64x64 / 128x128 kernels are bound
// by their staging chain (global load -> convert -> ds_write -> barrier, ONE k-tile ahead): tools/probes/
// bf16_resident_gemm.hip measured 12 us for 5120x512x512 where they take 22.  Here:
//   * 4 waves (2 x 2), workgroup tile 128 x 128, wave tile 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16 (a = P
//     fragment, b = Q fragment: the accumulator map of gemm_bf16_wide.hpp, epilogues of gemm_f32.hpp reused), BK = 32;
//   * three LDS images of 24 KB (Q [128][32] fp32, P [128][32] bf16; the swizzles of gemm_bf16_wide.hpp, applied to the
//     DMA's per-lane SOURCE address): tile t + 2 is requested while tile t is multiplied, so a short reduce loop pays the
//     memory latency about once, not once per tile; ONE barrier per k-tile (the image a request overwrites was read two
//     barriers ago); the requests are spread behind the matrix instructions (back to back they queue up behind the CU's
//     address unit);
//   * two workgroups per CU (72 KB each).  With BK = 64 and one workgroup per CU (144 KB) the launches of the decoder layers
//     (160 .. 960 tiles) ran 1.9 .. 3.75 rounds instead of 1 .. 2: 215 TF where the register-staged kernels reach 360.
#pragma once
#include "gemm_bf16.hpp"

namespace dma {
constexpr int HALVES = 1;                                          // 32-wide image k-tiles per LDS image
constexpr int BM = 128, BN = 128, BK = 32 * HALVES, NT = 2, MT = 2, NSTAGE = 3;
constexpr int Q_HALF = BM * 128, P_HALF = BN * 64;                 // bytes: [rows][32 fp32], [rows][32 bf16]
constexpr int Q_BYTES = HALVES * Q_HALF, P_BYTES = HALVES * P_HALF, STAGE = Q_BYTES + P_BYTES;
constexpr int LDS_BYTES = NSTAGE * STAGE;                          // 72 KB: two workgroups per CU
constexpr int QREQ = 4, PREQ = 2;                                  // DMA requests per wave per 32-wide half tile
constexpr int REQ = HALVES * (QREQ + PREQ);                        // per thread per k-tile
constexpr int KSTEPS = 2 * HALVES;                                 // 16-wide matrix-instruction steps per k-tile

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ int fq(int row) { return (row >> 1) & 7; }      // 16-byte chunk swizzle of the 128-byte fp32 rows

template <class Epi>
__device__ __forceinline__ void gemm_dma_body(const GemmShape& g, const Epi& epi, const int bid, const int z) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const int nk = kend > kbeg ? (kend - kbeg) / BK : 0;          // the launcher guarantees whole k-tiles

  // ---- DMA sources.  Q: request j of this wave covers rows (wave*4 + j)*8 .. +7 of a half tile, lane -> (row, chunk
  //      position); P: request j covers rows (wave*2 + j)*16 .. +15 of one 32-wide k-tile of the image
  const char* qsrc[QREQ];
  const char* psrc[PREQ];
#pragma unroll
  for (int j = 0; j < QREQ; ++j) {
    const int R = (wave * QREQ + j) * 8 + (lane >> 3);
    const int gc = min(m0 + R, Meff - 1);
    const int ph = g.rows ? g.rows[gc] : gc;
    qsrc[j] = reinterpret_cast<const char*>(g.Q + (int64_t)ph * g.ldq + kbeg) + 16 * ((lane & 7) ^ fq(R));
  }
#pragma unroll
  for (int j = 0; j < PREQ; ++j) {
    const int Rp = (wave * PREQ + j) * 16 + (lane >> 2);
    const int gn = min(n0 + Rp, g.N - 1);
    psrc[j] = reinterpret_cast<const char*>(g.Pimg + (int64_t)(kbeg >> 5) * g.ldpi) + (int64_t)gn * 64 + 16 * ((lane & 3) ^ lds_sw(Rp));
  }
  const int64_t p_tile_bytes = g.ldpi * 2;
  // request i (of REQ) of tile kt into image `stage`: i = h * 6 + (0..3: Q request j | 4..5: P request j - 4)
  auto issue_one = [&](int kt, int stage, int i) __attribute__((always_inline)) {
    unsigned char* base = smem_raw + stage * STAGE;
    const int h = i / (QREQ + PREQ), j = i % (QREQ + PREQ);
    if (j < QREQ)
      __builtin_amdgcn_global_load_lds(qsrc[j] + (int64_t)kt * (BK * 4) + h * 128, (lds_ptr)(base + h * Q_HALF + (wave * QREQ + j) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(psrc[j - QREQ] + (int64_t)(HALVES * kt + h) * p_tile_bytes,
                                       (lds_ptr)(base + Q_BYTES + h * P_HALF + (wave * PREQ + j - QREQ) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < REQ; ++i) issue_one(kt, stage, i);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  if (nk > 0) issue(0, 0);
  if (nk > 1) issue(1, 1);
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (tile kt + 1 may still be in flight)
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(REQ == 6, "the wait above counts the requests of one tile");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // tile kt + 2 goes into the image tile kt - 1 was read from (everybody is past it): its requests are spread over the
    // four k-steps below, three behind each step's matrix instructions -- issued back to back, the four waves queue up
    // behind the CU's address unit (~80 cycles per request) before any of them starts on the fragments
    const bool more = kt + 2 < nk;
    const int nst = st == 0 ? 2 : st - 1;
    const unsigned char* sq = smem_raw + st * STAGE;
    const unsigned char* sp = sq + Q_BYTES;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int h = s >> 1, s2 = s & 1;
      bf16x8_t fqv[MT], fpv[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const int row = wm * 64 + a * 32 + l31;
        const unsigned char* rp = sq + h * Q_HALF + row * 128;
        const int c0 = (4 * s2 + 2 * lh) ^ fq(row);
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp + 16 * c0);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 16 * (c0 ^ 1));
        const u32x4 w = {bf16_pack_rne(r0[0], r0[1]), bf16_pack_rne(r0[2], r0[3]), bf16_pack_rne(r1[0], r1[1]), bf16_pack_rne(r1[2], r1[3])};
        fqv[a] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int row = wn * 64 + b * 32 + l31;
        fpv[b] = *reinterpret_cast<const bf16x8_t*>(sp + h * P_HALF + row * 64 + 16 * ((2 * s2 + lh) ^ lds_sw(row)));
      }
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fpv[b], fqv[a], acc[a][b], 0, 0, 0);
      if (more) {
#pragma unroll
        for (int i = 0; i < REQ / KSTEPS; ++i) issue_one(kt + 2, nst, s * (REQ / KSTEPS) + i);
      }
    }
    st = st == 2 ? 0 : st + 1;
  }

  // ---- accumulator map (a = P fragment, b = Q fragment): register r of acc[a][b] is row (m) a*32 + l31, column (n)
  //      b*32 + 8*(r >> 2) + 4*lh + (r & 3): four consecutive columns per register quad, handed to the epilogues of
  //      gemm_f32.hpp as 8 float4s per row block (as gemm_bf16_wide.hpp does)
  constexpr int TN = NT * 4;
  int ncol[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) ncol[b] = n0 + wn * 64 + (b >> 2) * 32 + (b & 3) * 8 + 4 * lh;
  const bool fast = epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    f32x4 acc4[1][TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
      acc4[0][b] = f32x4{acc[a][b >> 2][4 * (b & 3)], acc[a][b >> 2][4 * (b & 3) + 1], acc[a][b >> 2][4 * (b & 3) + 2], acc[a][b >> 2][4 * (b & 3) + 3]};
    int mlog[1] = {m0 + wm * 64 + a * 32 + l31};
    int mphys[1] = {(g.rows && mlog[0] < Meff) ? g.rows[mlog[0]] : mlog[0]};
    if (fast) epi.template tile_fast<1, TN, true>(acc4, mphys, ncol, g.N, z);
    else epilogue_all<0, 1, TN, true, Epi>(epi, acc4, mlog, mphys, ncol, Meff, g.N, z);
  }
}

template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(GemmShape g, Epi epi) {
  gemm_dma_body<Epi>(g, epi, (int)blockIdx.x, (int)blockIdx.z);
}
}  // namespace dma
