// The 64 x 64 / 32 x 64 member of the bf16 matrix-core GEMM family (gemm_bf16.hpp) for the THROUGHPUT mode's launches of a few hundred
// tiles -- the decoder layers' nn.Linear forward and dX (models/bert.py:139-247) at 1-5 k rows, N, K = 512 .. 2048 -- built around the
// load LATENCY instead of the load bandwidth:
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]      Qop: fp32, k-contiguous (activations / dZ), live-row list;  Pop: pre-split bf16 image
//
//   * both operands go global -> LDS by DMA (buffer_load_dwordx4 ... lds): no register staging, no LDS stores, and the requests of
//     NSTAGE - 1 stages are in flight while one is consumed.  gemm_bf16_kernel<64, 64> holds a k-tile in registers between its
//     load and its LDS store, so at most two k-tiles are in flight and a launch of ~1 workgroup per SIMD runs at one memory
//     latency per k-tile (16 of them for K = 512: 14 us of a 20 us launch).
//   * every per-lane address is k-invariant (the row of the live-row list and the swizzled 16-byte chunk): computed once; the k-tile
//     advances through the request's SCALAR offset -- a stage costs its DMA instructions and nothing else.
//   * Qop lands as raw fp32 ([64 rows][32 k] = 128-byte rows per 32-deep sub-tile, chunk c of row r at c ^ qsw(r): conflict-free
//     ds_read_b128 under the instruction's lane groups, MI355X_MICROARCH.md "LDS") and is rounded to bf16 on the FRAGMENTS, in
//     registers (bf16_split2<1>: the values the staging of gemm_bf16.hpp produces, bit for bit); Pop is a pure copy of the
//     k-tile-major image ([64 rows][32 bf16], lds_sw as everywhere).
//   * one barrier per stage: it publishes the stage the waves have just waited for (counted vmcnt) and frees the buffer the
//     previous stage was read from, which the next DMA then refills.
//   * accumulator map and epilogues (EpiLinear / EpiStore / EpiArgmax) are those of gemm_bf16_kernel: the results are the same bits.
//   Measured inside the captured steps (profiles/r05_dma64_timeline.txt): launches of 16 .. 640 tiles gain 1-12 us each (NAB B = 64:
//   1.100 -> 1.017 ms per step; NACF B = 128 bf16: 1.74 -> 1.69 ms), launches of 1100+ tiles lose 2-11 us (two workgroups per CU
//   against three; every wave converts its own fragments) and stay on gemm_bf16_kernel.  The exact mode (three planes, six terms:
//   its k-tile is matrix-bound, not latency-bound) tied or lost with the same structure (2.556 vs 2.555-2.570 ms) and is not built.
#pragma once
#include "gemm_bf16.hpp"

namespace dma64 {

constexpr int BN = 64;      // BM = 64 or 32 (template): the 32-row tile doubles the workgroups of a launch that is under ~2 tiles per CU
typedef __attribute__((address_space(3))) void* lds_ptr;

// stage geometry: KS sub-tiles of 32 k per stage, NSTAGE stages in LDS
template <int BM> struct Geo {
  static constexpr int NS = 1;
  static constexpr int KS = 2;
  static constexpr int NSTAGE = 3;
  static constexpr int A_SUB = BM * 128;                    // [BM rows][32 fp32]
  static constexpr int AREQ = BM / 32;                      // DMA requests per wave for one sub-tile of Qop (8 rows each)
  static constexpr int B_PLANE = BN * 64;                   // [64 rows][32 bf16]
  static constexpr int B_SUB = NS * B_PLANE;
  static constexpr int STAGE = KS * (A_SUB + B_SUB);        // BM = 64: 24 KiB, BM = 32: 16 KiB
  static constexpr int LDS_BYTES = NSTAGE * STAGE;          // 72 KiB / 48 KiB: two / three workgroups per CU
  static constexpr int REQ = KS * (AREQ + NS);              // DMA requests per wave and stage
};

// chunk swizzle of the fp32 image (8 chunks of 16 bytes per 128-byte row): a fragment lane (row i = lane & 15, g = lane >> 4) reads
// chunks 2g and 2g + 1; with this XOR each of the instruction's four 16-lane groups covers all 64 banks once
__device__ __forceinline__ int qsw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_dma64_kernel(GemmShape g, Epi epi) {
  using G = Geo<BM>;
  constexpr int NS = G::NS, KS = G::KS, NSTAGE = G::NSTAGE, TM = BM / 32, TN = 2, WTM = BM / 2, AREQ = G::AREQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 15, lg = lane >> 4;
  const int bid = blockIdx.x, z = blockIdx.z;

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
  const int kbeg = z * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
  const int nsub = kend > kbeg ? (kend - kbeg) >> 5 : 0;      // (K and k_per_split are multiples of 32: the launcher checks)
  const int nstep = (nsub + KS - 1) / KS;

  // ---- descriptors and the k-invariant per-lane offsets
  const uint32_t a_bytes = (uint32_t)min((uint64_t)g.M * (uint64_t)g.ldq * 4u, (uint64_t)0x7ffffff0u);
  const uint32_t kt_bytes = (uint32_t)(g.ldpi * 2), plane_bytes = (uint32_t)(g.pimg_plane * 2);
  const uint32_t b_bytes = (uint32_t)((g.K + 31) >> 5) * kt_bytes + (NS - 1) * plane_bytes;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.Q, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.Pimg, 0, b_bytes, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;      // past either extent: the request returns zeros
  uint32_t voa[2], vob;      // (AREQ entries used.  Sized by the template-dependent constant, the array captured by the lambda below
                             //  makes hipcc drop the kernel's HOST stub without a diagnostic: undefined symbol at load time)
#pragma unroll
  for (int j = 0; j < AREQ; ++j) {
    const int r = wave * (8 * AREQ) + j * 8 + (lane >> 3), m = m0 + r;
    const int ph = m < Meff ? (g.rows ? g.rows[m] : m) : -1;
    voa[j] = ph >= 0 ? (uint32_t)ph * (uint32_t)(g.ldq * 4) + (uint32_t)(((lane & 7) ^ qsw(r & 15)) << 4) : OOB;
  }
  {
    const int r = wave * 16 + (lane >> 2);
    vob = (uint32_t)min(n0 + r, g.N - 1) * 64u + (uint32_t)(((lane & 3) ^ lds_sw(r)) << 4);
  }
  // fragment addresses inside a sub-tile
  uint32_t fa[TM][2], fb[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e) fa[a][e] = (uint32_t)((wm * WTM + a * 16 + li) * 128 + (((2 * lg + e) ^ qsw(li)) << 4));
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int row = wn * 32 + b * 16 + li;
    fb[b] = (uint32_t)(KS * G::A_SUB + row * 64 + ((lg ^ lds_sw(row)) << 4));
  }

  auto issue = [&](int step, int buf) {      // every request of one stage (wave-uniform arguments)
    unsigned char* base = smem + buf * G::STAGE;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int sub = step * KS + u;
      const bool live = sub < nsub;
      const uint32_t ka = (uint32_t)(kbeg + 32 * sub) * 4u;
      const uint32_t kb = (uint32_t)((kbeg >> 5) + sub) * kt_bytes;
#pragma unroll
      for (int j = 0; j < AREQ; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + u * G::A_SUB + (wave * (8 * AREQ) + j * 8) * 128), 16, live ? voa[j] : OOB, ka, 0,
                                                 0);
#pragma unroll
      for (int p = 0; p < NS; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(base + KS * G::A_SUB + (u * NS + p) * G::B_PLANE + wave * 16 * 64), 16,
                                                 live ? vob : OOB, kb + p * plane_bytes, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * G::STAGE;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      bf16x8_t pf[TN][NS], qf[TM][NS];
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int p = 0; p < NS; ++p) pf[b][p] = *reinterpret_cast<const bf16x8_t*>(base + fb[b] + (u * NS + p) * G::B_PLANE);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(base + u * G::A_SUB + fa[a][0]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(base + u * G::A_SUB + fa[a][1]);
        u32x4 pl[NS];
        uint32_t w[NS];
        bf16_split2<NS>(lo[0], lo[1], w);
#pragma unroll
        for (int p = 0; p < NS; ++p) pl[p][0] = w[p];
        bf16_split2<NS>(lo[2], lo[3], w);
#pragma unroll
        for (int p = 0; p < NS; ++p) pl[p][1] = w[p];
        bf16_split2<NS>(hi[0], hi[1], w);
#pragma unroll
        for (int p = 0; p < NS; ++p) pl[p][2] = w[p];
        bf16_split2<NS>(hi[2], hi[3], w);
#pragma unroll
        for (int p = 0; p < NS; ++p) pl[p][3] = w[p];
#pragma unroll
        for (int p = 0; p < NS; ++p) qf[a][p] = __builtin_bit_cast(bf16x8_t, pl[p]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[b][0], qf[a][0], acc[a][b], 0, 0, 0);
    }
  };

  // ---- pipeline
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nstep) issue(s, s);
  int buf = 0, nbuf = NSTAGE - 1;
  for (int s = 0; s < nstep; ++s) {
    const int ahead = min(NSTAGE - 2, nstep - 1 - s);      // stages requested after this one
    if (ahead >= 2) wait_vm<2 * G::REQ>();
    else if (ahead == 1) wait_vm<G::REQ>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NSTAGE - 1 < nstep) issue(s + NSTAGE - 1, nbuf);
    compute(buf);
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    nbuf = nbuf + 1 == NSTAGE ? 0 : nbuf + 1;
  }

  // ---- epilogue (accumulator map = the KC / KC map of gemm_f32.hpp)
  int mlog[TM], ncol[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) mlog[a] = m0 + wm * WTM + a * 16 + li;
#pragma unroll
  for (int j = 0; j < TN; ++j) ncol[j] = n0 + wn * 32 + j * 16 + lg * 4;
  if constexpr (!Epi::kArgmax) {
    int mphys[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) mphys[a] = (g.rows && mlog[a] < Meff) ? g.rows[mlog[a]] : mlog[a];
    if (epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N) epi.template tile_fast<TM, TN, true>(acc, mphys, ncol, g.N, z);
    else epilogue_all<0, TM, TN, true, Epi>(epi, acc, mlog, mphys, ncol, Meff, g.N, z);
  } else {
    __syncthreads();      // the last stage's fragments are read: the scratch may reuse the images
    argmax_epilogue<BM, BN, 2, 2, TM, TN>(reinterpret_cast<float*>(smem), g, epi, acc, m0, n0, Meff, tile_n, wm, wn, li, lg, tid);
  }
}

// host side: eligibility and launch
inline bool eligible(const GemmShape& g, int splits) {
  if (!g.Pimg || (g.K & 31) != 0) return false;
  if (splits > 1 && (g.k_per_split & 31) != 0) return false;
  if ((uint64_t)g.M * (uint64_t)g.ldq * 4u >= 0x7ffffff0ull) return false;                          // 31-bit request offsets
  const uint64_t bb = (uint64_t)((g.K + 31) >> 5) * (uint64_t)g.ldpi * 2u;
  return bb < 0x7ffffff0ull && (uint64_t)g.N * 64u < 0x7ffffff0ull;
}
template <int BM, class Epi>
inline void launch(const GemmShape& g, const Epi& epi, dim3 grid, hipStream_t s) {
  auto kern = gemm_dma64_kernel<BM, Epi>;
  constexpr int lds = Geo<BM>::LDS_BYTES;
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, epi);
}

}  // namespace dma64
