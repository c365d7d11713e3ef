// The weight-gradient member of the wide-wave-tile family (gemm_bf16_wide.hpp): the same one-workgroup-per-CU geometry,
// matrix instruction, fixed accumulation registers and exact three-way split, for the TN product whose operands are both
// reduce-dimension-major activations:
//
//   C[i][j] = sum_m A[m][i] * B[m][j]        A = dZ [M, N] fp32, B = X [M, K] fp32 (nn.Linear: dW = dZ^T X), m optionally
//                                            through a live-row list; split over m into slabs (EpiStore)
//
//   * workgroup tile 128 (i) x 256 (j), 4 waves as 2 x 2, wave tile 64 x 128, v_mfma_f32_32x32x16_bf16 issued with
//     a = B fragment, b = A fragment -- the accumulator map and the epilogue are those of gemm_wide_body.
//   * neither operand can come by DMA (an LDS image of a DMA is lane-linear, the matrix instruction wants 8 consecutive m
//     per lane): both are staged through registers by the transposing stager of the 128 x 128 kernel (StageF32MC: a thread
//     loads 4 m-rows x 4 columns, splits, and writes one 8-byte piece per column and plane) into k-contiguous split planes.
//     One k-tile is three such sub-images of [3 planes][128 rows][32 bf16] = 24 KB: A | B rows 0..127 | B rows 128..255;
//     two stages = 144 KB.
//   * per k-tile and wave: 96 matrix instructions (3072 cycles), 18 fragment reads, and the staging of the NEXT tile --
//     168 split steps (264 vector instructions), 36 ds_write_b64, 12 global loads of the tile after -- spread over the
//     first 72 slots; the barrier sits before the last column block, whose slots carry the first fragment reads of the next
//     tile (as in the wide forward kernel).  The A fragments live in ONE register set: the (k-step 0) half is re-read for
//     the next tile while the last block's k-step-1 instructions run, the other half while the next tile's k-step-0 ones do.
//   * vs the 128 x 128 kernel (2 workgroups per CU, 64 x 64 wave tiles): 36 KB of fragment reads + 18 KB of stores per
//     wave per 96 matrix instructions instead of 72 + 36 KB -- that kernel is bound by exactly those.
#pragma once
#include "gemm_bf16_wide.hpp"

namespace widet {
using wide::IC;
using wide::for_each_ic;
using wide::lds_read;
using wide::f32x16;
using wide::mfma_fixed;
using wide::acc_zero_all;
using wide::acc_read;
constexpr int TM = 128, TN = 256, BK = 32;
constexpr int PLANE = 128 * 64;            // [128 rows][32 bf16]
constexpr int SUB = 3 * PLANE;             // one operand sub-image (three planes): 24576 bytes
constexpr int STAGE = 3 * SUB;             // A | B low | B high
constexpr int LDS_BYTES = 2 * STAGE;       // 147456
constexpr int NT = 4, NSLOT = 96, PH = 24;
constexpr bool MANUAL_VM = false;
#ifndef WIDET_ABL
#define WIDET_ABL 0      // tuning builds: 1 = no split, 2 = no stores, 4 = no loads in the loop, 8 = no fragment reads
#endif

typedef StageF32MC<128, 3, 0> Stager;

template <int OFF, class V>
__device__ __forceinline__ void lds_read_sync(V& r, uint32_t addr) {        // read AND wait: the result is there when this returns
  asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "n"(OFF));
}
// one float4 row of the stager's 4 x 4 unit (the loads of a tile are spread over four slots)
template <bool FAST>
__device__ __forceinline__ void load_one(Stager& st, int j, int k0, int kend, int64_t ld, const int* __restrict__ kmap) {
  const int gk = k0 + 4 * st.mq + j;
  if constexpr (FAST) {
    const int pk = kmap ? st.kidx[j] : gk;
    st.v[j] = *reinterpret_cast<const f32x4*>(st.ptr + (int64_t)pk * ld);
  } else {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (gk < kend && st.nvalid > 0) {
      const int pk = kmap ? kmap[gk] : gk;
      const float* p = st.ptr + (int64_t)pk * ld;
      if (st.nvalid == 4) t = *reinterpret_cast<const f32x4*>(p);
      else {
        t[0] = p[0];
        if (st.nvalid > 1) t[1] = p[1];
        if (st.nvalid > 2) t[2] = p[2];
      }
    }
    st.v[j] = t;
  }
}
// The steady loop's global loads are inline assembly with hand-counted waits: left to the compiler, the loop header waits
// for vmcnt(0) (its counter model merges the entry and the back edge conservatively) -- every trip then sat out a full
// memory round trip for loads issued 20 slots earlier (measured: 6300 cycles per k-tile for 3072 of matrix work).
// FAST loads: wave-uniform base (scalar registers, advanced per tile) + a 32-bit byte offset per lane that does not change
// from tile to tile (no 64-bit address arithmetic in the loop).  With a row list the offset is index * pitch (32-bit: the
// launcher keeps problems whose operands span 4 GB or more on the 128 x 128 kernel).
template <bool KMAP>
__device__ __forceinline__ void load_one_asm(Stager& st, int j, const char* tile_base, uint32_t lane_off, uint32_t ld_bytes) {
  const uint32_t off = KMAP ? (uint32_t)st.kidx[j] * ld_bytes + lane_off : lane_off;
  st.v[j] = *reinterpret_cast<const f32x4*>(tile_base + off);
}
__device__ __forceinline__ void load_kidx_asm(Stager& st, int k0, int kmax, const int* __restrict__ kmap) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    st.kidx[j] = kmap[min(k0 + 4 * st.mq + j, kmax - 1)];
  }
}
// all but the N newest vector-memory loads have returned; `data` rides along so that no use of it moves above the wait,
// `idx` (the row indices of an OLDER request) likewise
template <int N>
__device__ __forceinline__ void wait_vm(Stager& data, Stager& idx) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(data.v[0]), "+v"(data.v[1]), "+v"(data.v[2]), "+v"(data.v[3]),
               "+v"(idx.kidx[0]), "+v"(idx.kidx[1]), "+v"(idx.kidx[2]), "+v"(idx.kidx[3]) : "n"(N));
}
// tell the compiler's own wait-count model that these registers are needed NOW (it inserts its waits here, not in the loop)
__device__ __forceinline__ void settle(Stager& st) {
  asm volatile("" : "+v"(st.v[0]), "+v"(st.v[1]), "+v"(st.v[2]), "+v"(st.v[3]), "+v"(st.kidx[0]), "+v"(st.kidx[1]), "+v"(st.kidx[2]),
               "+v"(st.kidx[3]));
}
// Row r of a sub-image sits in row slot rho(r) = r ^ ((r >> 2) & 1).  A store instruction writes column e of every lane's
// unit, i.e. rows 4 rq + e of 8 consecutive row quads: 256 bytes apart, all on the SAME half of the LDS banks (a 64-byte
// row covers 16 of the 32 store banks).  With rho the odd row quads swap their row pairs, so one instruction alternates
// between the two halves (measured: the stores were the largest single item of the k-loop, 0.25 of 1.23 ms).  The swizzle of
// the 16-byte chunks depends on r >> 2 only and is untouched; a fragment read's 16-lane groups still cover the same set
// of slots (the permutation stays inside a group of 4 rows).
#ifndef WIDET_NO_RHO
__device__ __forceinline__ int rho(int r) { return r ^ ((r >> 2) & 1); }
#else
__device__ __forceinline__ int rho(int r) { return r; }
#endif
// one of the stager's 12 stores: column e of its unit, plane p (layout of StageF32MC::store, rows through rho)
__device__ __forceinline__ void store_one(const Stager& st, unsigned char* sub, int e, int p) {
  u32x2* p8 = reinterpret_cast<u32x2*>(sub);
  const int c = st.mq >> 1, half = st.mq & 1;
  const int row = 4 * st.rq + e;
  p8[(p * (128 * 4) + rho(row) * 4 + (c ^ lds_sw(row))) * 2 + half] = st.w[e][p];
}
__device__ __forceinline__ void write_all(Stager& st, unsigned char* sub) {
  st.split();
#pragma unroll
  for (int q = 0; q < 12; ++q) store_one(st, sub, q / 3, q % 3);
}

__device__ __forceinline__ void gemm_widet_body(const GemmShape& g, const EpiStore& epi, const int tile, const int z, const int nz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  int Keff = g.K;
  if (g.count) Keff = min(Keff, *g.count);
  const int tiles_n = (g.N + TN - 1) / TN;
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int m0 = tile_m * TM, n0 = tile_n * TN;
  int kps = g.k_per_split;
  if (g.count && nz > 1) {
    kps = (((Keff + nz - 1) / nz) + BK - 1) / BK * BK;   // re-balance over the LIVE rows
    if (kps < BK) kps = BK;
  }
  const int kbeg = z * kps;
  const int kend = min(Keff, kbeg + kps);
  const int nk = __builtin_amdgcn_readfirstlane(kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0);
  const int* kmap = g.rows;
  const bool rows_full = (m0 + TM <= g.M) && (n0 + TN <= g.N);
  const int64_t lda = g.ldq, ldb = g.ldp;

  Stager sa, sb0, sb1;
  sa.init(g.Q, m0, g.M, tid);
  sb0.init(g.P, n0, g.N, tid);
  sb1.init(g.P, n0 + 128, g.N, tid);
  f32x4 qsum = {0.f, 0.f, 0.f, 0.f};      // this thread's column sums of A (the bias gradient), all k-tiles
  // byte offsets of this thread's 4 columns inside a row of each operand (FAST loads)
  const uint32_t coff_a = (uint32_t)(m0 + 4 * sa.rq) * 4u, coff_b0 = (uint32_t)(n0 + 4 * sb0.rq) * 4u, coff_b1 = (uint32_t)(n0 + 128 + 4 * sb1.rq) * 4u;

  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  // fragment addresses: row r of a sub-image, 16-byte chunk (2 s + lh) ^ sw(r) of the plane's 64-byte row
  //   B (a operand): sub-image 1 + wn, row nt * 32 + l31;   A (b operand): sub-image 0, row wm * 64 + mt * 32 + l31
  const uint32_t sw = lds_sw(l31);
  const int l31p = rho(l31);
  const uint32_t b_rd0 = lds0 + (1 + wn) * SUB + l31p * 64 + ((lh ^ sw) * 16);         // s = 0; + stage, plane, nt * 2048
  const uint32_t b_rd[2][2] = {{b_rd0, b_rd0 ^ 32}, {b_rd0 + STAGE, (b_rd0 ^ 32) + STAGE}};      // [stage][s]
  const uint32_t a_rd0 = lds0 + (wm * 64 + l31p) * 64 + ((lh ^ sw) * 16);              // + stage, plane, mt * 2048
  const uint32_t a_rd[2][2] = {{a_rd0, a_rd0 ^ 32}, {a_rd0 + STAGE, (a_rd0 ^ 32) + STAGE}};
  static_assert(PLANE * 2 + 3 * 2048 < 65536, "ds_read offsets are 16 bits");

  u32x4 qf[2][2][3];                  // A fragments [mt][s][plane]
  u32x4 pf[2][2][3];                  // B fragments of one column block [ping-pong][s][plane]
  auto read_pf_one = [&](auto set_c, auto stg_c, auto nt_c, auto i_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, STG = decltype(stg_c)::value, NTI = decltype(nt_c)::value, I = decltype(i_c)::value;
    constexpr int S = I / 3, PL = I % 3;
    lds_read<PL * PLANE + NTI * 2048>(pf[SET][S][PL], b_rd[STG][S]);
  };
  auto read_qf_one = [&](auto stg_c, auto s_c, auto i_c) __attribute__((always_inline)) {      // I = mt * 3 + plane
    constexpr int STG = decltype(stg_c)::value, S = decltype(s_c)::value, I = decltype(i_c)::value;
    constexpr int MTI = I / 3, PL = I % 3;
    lds_read<PL * PLANE + MTI * 2048>(qf[MTI][S][PL], a_rd[STG][S]);
  };
  constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};     // six cross terms, smallest first

  // ---- one k-tile read from stage PAR; the tile after it is split from the stagers' registers into stage PAR ^ 1 and the
  //      one after that is loaded into them.  Slot = one matrix instruction + its share of everything else:
  //        stager S (A, B low, B high): split steps (3 per slot) in slots 19 S .. 19 S + 18, its 12 stores (1 per slot) in the
  //        12 slots behind them, then the 4 loads of the tile after next + the row indices of the one after that
  //        (B high: behind the barrier, slots 73..77);  barrier at slot 72 (all stores done);
  //        B fragments of the next column block in the first 6 slots of a block (block 0 of the NEXT tile in block 3);
  //        A fragments: k-step 1 of THIS tile in slots 0..5, k-step 0 of the NEXT tile in slots 84..89
  //      FAST: loads by inline assembly; the vector-memory queue then holds, oldest first, per tile call
  //        [A: 4 loads, 4k row indices] [B low: 4, 4k] [B high: 4, 4k]   (k = 1 with a row list), hence the wait counts below
  auto tile_fn = [&](auto par_c, auto fast_c, auto kmap_c, const int kt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FAST = decltype(fast_c)::value;
    constexpr bool KMAP = decltype(kmap_c)::value;
    constexpr int KQ = KMAP ? 4 : 0;
    // uniform bases of the tile this call loads (no row list: row k2 of each operand; row list: the operand itself)
    const char* const base_a = reinterpret_cast<const char*>(g.Q) + (KMAP ? (int64_t)0 : (int64_t)(kbeg + (kt + 2) * BK) * lda * 4);
    const char* const base_b = reinterpret_cast<const char*>(g.P) + (KMAP ? (int64_t)0 : (int64_t)(kbeg + (kt + 2) * BK) * ldb * 4);
    unsigned char* const wr = smem_raw + (PAR ^ 1) * STAGE;
    const int k2 = kbeg + (kt + 2) * BK;        // first reduce index of the tile loaded in this call
    auto slot = [&](auto gs_c) __attribute__((always_inline)) {
      constexpr int gs = decltype(gs_c)::value;
      constexpr int p = gs / PH, sl = gs % PH, s = sl / 12, term = (sl / 2) % 6, mt = sl % 2;
      if constexpr (sl == 0 || gs == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments of this block / k-step
      if constexpr (gs == 3 * PH) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (also: every store of the next tile has landed)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      mfma_fixed<mt * 4 + p>(pf[p & 1][s][TP[term]], qf[mt][s][TQ[term]]);
      // ---- fragment reads
      if constexpr (sl < 6 && !(WIDET_ABL & 8)) {
        if constexpr (p < 3) read_pf_one(IC<(p + 1) & 1>{}, IC<PAR>{}, IC<p + 1>{}, IC<sl>{});
        else read_pf_one(IC<0>{}, IC<PAR ^ 1>{}, IC<0>{}, IC<sl>{});
      }
      if constexpr (gs < 6 && !(WIDET_ABL & 8)) read_qf_one(IC<PAR>{}, IC<1>{}, IC<gs>{});
      if constexpr (gs >= 84 && gs < 90 && !(WIDET_ABL & 8)) read_qf_one(IC<PAR ^ 1>{}, IC<0>{}, IC<gs - 84>{});
      // ---- staging of the next tile
      for_each_ic([&](auto st_c) __attribute__((always_inline)) {
        constexpr int S = decltype(st_c)::value;
        Stager& st = S == 0 ? sa : (S == 1 ? sb0 : sb1);
        constexpr int g0 = 19 * S;
        if constexpr (FAST && MANUAL_VM) {
          // this stager's data (requested one call ago) has returned -- and with it the row indices the stager before it
          // asked for; younger requests stay in flight
          if constexpr (S == 0 && gs == 0) wait_vm<8 + 3 * KQ>(sa, sa);
          if constexpr (S == 1 && gs == g0) wait_vm<4 + 2 * KQ>(sb0, sa);
          if constexpr (S == 2 && gs == g0) wait_vm<4 + 2 * KQ + KQ>(sb1, sb0);
          if constexpr (S == 2 && gs == 73 && KMAP) wait_vm<8 + 2 * KQ>(sb1, sb1);
        }
        if constexpr (S == 0 && gs == 0) qsum += (st.v[0] + st.v[1]) + (st.v[2] + st.v[3]);
        if constexpr (gs >= g0 && gs < g0 + 19) {
          for_each_ic([&](auto i_c) __attribute__((always_inline)) {
            constexpr int idx = (gs - g0) * 3 + decltype(i_c)::value;
            if constexpr (idx < 56 && !(WIDET_ABL & 1)) st.split_step(idx % 8, idx / 8);
          }, std::make_integer_sequence<int, 3>{});
        }
        if constexpr (gs >= g0 + 19 && gs < g0 + 31) {
          constexpr int q = gs - (g0 + 19);
          if constexpr (!(WIDET_ABL & 2)) store_one(st, wr + S * SUB, q / 3, q % 3);
        }
        constexpr int l0 = S == 2 ? 73 : g0 + 31;
        if constexpr (gs >= l0 && gs < l0 + 4 && !(WIDET_ABL & 4)) {
          if constexpr (FAST) {
            constexpr int J = gs - l0;
            const uint32_t ldB = (uint32_t)(S == 0 ? lda : ldb) * 4u;
            const uint32_t col = (S == 0 ? coff_a : (S == 1 ? coff_b0 : coff_b1));
            load_one_asm<KMAP>(st, J, S == 0 ? base_a : base_b, KMAP ? col : col + (uint32_t)(4 * st.mq + J) * ldB, ldB);
          }
          else load_one<false>(st, gs - l0, k2, kend, S == 0 ? lda : ldb, kmap);
        }
        if constexpr (gs == l0 + 4 && FAST && KMAP) load_kidx_asm(st, k2 + BK, Keff, kmap);
      }, std::make_integer_sequence<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
    };
    for_each_ic(slot, std::make_integer_sequence<int, NSLOT>{});
  };

  // ---- prologue: tile 0 into stage 0 (nothing to hide behind), tile 1 into the registers
  acc_zero_all<8>();
  if (nk > 0) {
    sa.load_checked(kbeg, kend, lda, kmap);
    sb0.load_checked(kbeg, kend, ldb, kmap);
    sb1.load_checked(kbeg, kend, ldb, kmap);
    qsum += (sa.v[0] + sa.v[1]) + (sa.v[2] + sa.v[3]);
    write_all(sa, smem_raw);
    write_all(sb0, smem_raw + SUB);
    write_all(sb1, smem_raw + 2 * SUB);
    sa.load_checked(kbeg + BK, kend, lda, kmap);
    sb0.load_checked(kbeg + BK, kend, ldb, kmap);
    sb1.load_checked(kbeg + BK, kend, ldb, kmap);
    sa.load_kidx(kbeg + 2 * BK, Keff, kmap);
    sb0.load_kidx(kbeg + 2 * BK, Keff, kmap);
    sb1.load_kidx(kbeg + 2 * BK, Keff, kmap);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the first fragments, read SYNCHRONOUSLY (read + wait in one statement): register pressure peaks here (both operands'
    // next tile in flight), and a spill copy of a register whose ds_read has not returned yet would save garbage -- the
    // compiler does not know these statements are loads (tools/check_wide_hazards.py rejects such copies anywhere)
    for_each_ic([&](auto i_c) __attribute__((always_inline)) {
      constexpr int I = decltype(i_c)::value;
      lds_read_sync<(I % 3) * PLANE>(pf[0][I / 3][I % 3], b_rd[0][I / 3]);
      lds_read_sync<(I % 3) * PLANE + (I / 3) * 2048>(qf[I / 3][0][I % 3], a_rd[0][0]);
    }, std::make_integer_sequence<int, 6>{});
    // a tile call loads tile kt + 2: the unchecked loader serves it while that tile is whole and inside the matrix
    int kt = 0;
    const int n_fast = rows_full ? (kend - kbeg) / BK - 2 : 0;      // calls kt < n_fast load a whole tile
    settle(sa); settle(sb0); settle(sb1);     // (the compiler's waits for the prologue's loads land here)
    if (kmap) {
#pragma nounroll
      for (; kt + 1 < n_fast; kt += 2) {
        tile_fn(IC<0>{}, std::true_type{}, std::true_type{}, kt);
        tile_fn(IC<1>{}, std::true_type{}, std::true_type{}, kt + 1);
      }
    } else {
#pragma nounroll
      for (; kt + 1 < n_fast; kt += 2) {
        tile_fn(IC<0>{}, std::true_type{}, std::false_type{}, kt);
        tile_fn(IC<1>{}, std::true_type{}, std::false_type{}, kt + 1);
      }
    }
#pragma nounroll
    for (; kt < nk; kt += 2) {
      tile_fn(IC<0>{}, std::false_type{}, std::false_type{}, kt);
      if (kt + 1 < nk) tile_fn(IC<1>{}, std::false_type{}, std::false_type{}, kt + 1);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // the last matrix instruction has written its rows
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (reads / loads that ran past the last tile have returned)
  __syncthreads();      // the images are dead: the column sums fold through them

  // ---- db[i] = sum_m A[m][i] (tile_n == 0 only): the 8 lanes of a row quad sit in one wave (StageF32MC: unit = rq * 8 + mq)
  const bool do_colsum = (g.colsum_out || g.colsum_part) && tile_n == 0;
  if (do_colsum) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      qsum[0] += __shfl_xor(qsum[0], o, 64); qsum[1] += __shfl_xor(qsum[1], o, 64);
      qsum[2] += __shfl_xor(qsum[2], o, 64); qsum[3] += __shfl_xor(qsum[3], o, 64);
    }
    float* red = reinterpret_cast<float*>(smem_raw);
    if (sa.mq == 0) *reinterpret_cast<f32x4*>(&red[4 * sa.rq]) = qsum;
    __syncthreads();
    if (tid < TM) {
      const int m = m0 + tid;
      if (m < g.M) {
        const float t = red[tid];
        if (g.colsum_out) g.colsum_out[m] = (g.colsum_beta != 0.f) ? t + g.colsum_beta * g.colsum_out[m] : t;
        else g.colsum_part[(int64_t)z * g.M + m] = t;
      }
    }
  }

  // ---- epilogue: the accumulator map of gemm_wide_body (row = i, four consecutive j per register quad), one row block at a time
  constexpr int TNQ = NT * 4;
  int ncol[TNQ];
#pragma unroll
  for (int b = 0; b < TNQ; ++b) ncol[b] = n0 + wn * 128 + (b >> 2) * 32 + (b & 3) * 8 + 4 * lh;
  const bool fast = epi.fast_ok() && rows_full;
  for_each_ic([&](auto a_c) __attribute__((always_inline)) {
    constexpr int A = decltype(a_c)::value;
    f32x16 acc[NT];
    for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int I = decltype(i_c)::value; acc_read<A * 4 + I>(acc[I]); },
                std::make_integer_sequence<int, NT>{});
    f32x4 acc4[1][TNQ];
#pragma unroll
    for (int b = 0; b < TNQ; ++b)
      acc4[0][b] = f32x4{acc[b >> 2][4 * (b & 3)], acc[b >> 2][4 * (b & 3) + 1], acc[b >> 2][4 * (b & 3) + 2], acc[b >> 2][4 * (b & 3) + 3]};
    int mlog[1] = {m0 + wm * 64 + A * 32 + l31};
    int mphys[1] = {mlog[0]};
    if (fast) epi.template tile_fast<1, TNQ, true>(acc4, mphys, ncol, g.N, z);
    else epilogue_all<0, 1, TNQ, true, EpiStore>(epi, acc4, mlog, mphys, ncol, g.M, g.N, z);
  }, std::make_integer_sequence<int, 2>{});
}

// grouped launch: the problem table and split-major XCD order of gemm_bf16_group_kernel (order 1), gx[p] = output tiles
__global__ __launch_bounds__(256, 1) void gemm_widet_group_kernel(GemmGroup<EpiStore> t) {
  int p = 0;
#pragma unroll 1
  while (p + 1 < t.n && (int)blockIdx.x >= t.wg0[p + 1]) ++p;
  const int local = (int)blockIdx.x - t.wg0[p];
  const int gx = t.gx[p];
  const int total = gx * t.nz[p];
  const int xq = total >> 3, xr = total & 7;
  const int xcd = local & 7, slot = local >> 3;
  if (slot >= xq + (xcd < xr ? 1 : 0)) return;
  const int l = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
  const int z = l / gx;
  gemm_widet_body(t.g[p], t.e[p], l - z * gx, z, t.nz[p]);
}
}  // namespace widet
