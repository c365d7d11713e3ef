// The "panel" member of the bf16 matrix-core GEMM family: the kernel for the SKINNY launches of the decoder layer
// (models/bert.py:139-247: q|k|v, the two attention output projections, the cross-attention query, the FFN pair, and their
// dX twins) -- 2-9 k live rows x 512..2048 columns x K = 512..2048, i.e. 20-140 row panels of 64 rows.  The 64x64 / 128x128
// kernels run these at 0.2 of the exact mode's roof: half a round of workgroups whose 16 k-tiles sit between an LDS
// pipeline fill and an epilogue of about the same length.  This kernel removes the shared LDS
// pipeline AND the per-tile prologue instead of tuning them:
//
//   C[m][n] = sum_k Qop[m][k] * Pop[n][k]      Qop: fp32, k-contiguous rows (activations / dZ, optional live-row list)
//                                               Pop: FRAGMENT-MAJOR pre-split bf16 image of the weight matrix
//
//   * workgroup tile 32 MT (rows) x 32 NT (columns), 4 waves, ONE per SIMD, and the four waves split the REDUCE dimension:
//     wave w owns the 64-deep k-chunks 4 j + w.  Nothing in the k-loop is shared between waves: no barrier, no common
//     LDS image, every wave runs at its own pace; the four partial tiles are added (fixed order w = 0..3: run-to-run and
//     launch-shape independent) through LDS after the loop, where the fused epilogues (EpiLinear / EpiStore) run
//     on float4s with all their loads issued up front.
//   * weights go global -> REGISTERS, no LDS: the fragment-major image (nacf_wimage_desc.fimg) stores, per (16-deep
//     k-step, 32-column block, bf16 term), exactly the 1 KB a wave needs as the matrix instruction's first operand
//     (lane = (column & 31, k-half), 8 bf16 each), so a fragment is ONE fully coalesced global_load_dwordx4.  The
//     fragments live in accumulation registers a128.. (two stages, re-loaded two k-steps ahead right after their last
//     use) which the matrix instruction reads directly -- they never cost a vector register or an LDS byte.
//   * activations go global -> registers -> LDS into a wave-private ring of two 64-deep chunks (a chunk: [rows][64 fp32],
//     the 16-byte slots XOR-swizzled so that the two ds_read_b128 of a 32-row fragment and the staging ds_write_b128 are
//     conflict-free), are read as raw fp32 fragments one k-step ahead and split into their three bf16 terms in registers,
//     three vector instructions per matrix-instruction slot (the split of gemm_bf16_wide.hpp).  (Staged by LDS-DMA, as
//     in the wide kernel, a request costs the issuing wave ~100 cycles between matrix instructions: 16 per chunk made the
//     k-step 2030 cycles instead of 1650; a global_load_dwordx4 + ds_write_b128 pair costs about 20, and the kernel has the
//     registers: its weights live in the accumulation file.)
//   * PERSISTENT, and the operand stream does not stop at a tile boundary: a workgroup walks a contiguous run of tiles, and
//     "two k-steps ahead" (weights), "two chunks ahead" (activations) and "one step ahead" (the split) simply run on into
//     the NEXT tile (the next tile's live-row indices are fetched a whole tile ahead, one row per lane, and handed to the
//     DMA requests by ds_bpermute).  Measured with per-workgroup wall-clock stamps, a one-tile-per-workgroup version of
//     this kernel spent 6-7 us between the end of one tile and the first matrix instruction of the next (kernel arguments ->
//     live-row count -> row list -> 56 KB of requests per wave -> first data) around a 6.5 us k-loop that itself runs at
//     93 % of the matrix rate.  In the stream only the partial-sum epilogue sits between two tiles' matrix instructions.
//   * per k-step a wave issues MT x NT x 6 matrix instructions on 8 accumulators a0..a127 (fixed registers, as in the
//     wide kernel) and, in their shadow, NT x 3 weight loads, 2 MT LDS reads, 44 MT split operations, the chunk DMAs.
//     Per 32 MT x 128 tile and k-step: 12 KB of weights + 2 MT KB of activations per wave through L1 for
//     1536 MT/2 matrix cycles (47 B/clk/CU at MT = 2: under the 64 B/clk the L1 delivers).
//   LDS: 4 waves x 2 chunks x 32 MT x 256 B = 64 MT KB (the ring; its chunk that is free at a tile's end carries the partial sums).
#pragma once
#include <utility>
#include "gemm_bf16.hpp"

namespace panel {
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "gemm_bf16_panel_regs.inc"

template <int MT, int NT, int NS> struct Geo {
  static constexpr int BM = 32 * MT, BN = 32 * NT;
  static constexpr int SLOT = BM * 256;              // one 64-deep chunk of the wave's activations: [BM][64 fp32]
  static constexpr int WAVE_LDS = 2 * SLOT;
  static constexpr int LDS_BYTES = 4 * WAVE_LDS;     // (the partial sums of a tile go through the chunk of the ring that is free at its end)
  static constexpr int DREQ = BM / 4;                // DMA requests (1 KB) per chunk
  static constexpr int NTERM = NS == 3 ? 6 : 1;
  static constexpr int NSLOT = NT * NTERM * MT;      // matrix instructions per k-step
  static constexpr int NB = NT * NS;                 // weight fragments per k-step
  static_assert(LDS_BYTES <= 160 * 1024 && NT % 2 == 0, "LDS / two column blocks per epilogue pass");
  static_assert(2 * NB <= 24, "two stages of weight fragments in a128..a223");
  static_assert(3 * DREQ <= NSLOT, "slots for the DMA requests");
};

#ifdef PANEL_TRACE
// tuning builds: [tile][8] stamps of wave 0: 0 tile start, 2 k-loop end, 3 epilogue end (shader clock, per XCD: not comparable
// across workgroups); 4, 5: the 100 MHz wall clock at tile start / end; 6: XCC id; 7: workgroup
__device__ unsigned long long* g_panel_trace = nullptr;
#define PANEL_MARK(t, i) do { if (g_panel_trace && tid == 0) { g_panel_trace[(size_t)(t) * 8 + (i)] = __builtin_readcyclecounter(); \
  if ((i) == 0) { g_panel_trace[(size_t)(t) * 8 + 4] = __builtin_amdgcn_s_memrealtime(); g_panel_trace[(size_t)(t) * 8 + 6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); \
                  g_panel_trace[(size_t)(t) * 8 + 7] = bx; } \
  if ((i) == 3) g_panel_trace[(size_t)(t) * 8 + 5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#if PANEL_TRACE >= 2
#define PANEL_STEP_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PANEL_STEP_ACC(i, a, b) step_cyc[i] += (b) - (a)
#else
#define PANEL_STEP_T(var) do { } while (0)
#define PANEL_STEP_ACC(i, a, b) do { } while (0)
#endif
#else
#define PANEL_MARK(t, i) do { } while (0)
#define PANEL_STEP_T(var) do { } while (0)
#define PANEL_STEP_ACC(i, a, b) do { } while (0)
#endif
#ifndef PANEL_ABL
#define PANEL_ABL 0       // tuning builds: 1 = no activation loads / stores in the loop, 2 = no row bookkeeping, 4 = no split, 8 = no weight loads
#endif

template <int OFF, class V>
__device__ __forceinline__ void lds_read(V& r, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_write(uint32_t addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// 16 bytes per lane from base + voff into a staging register the compiler does not know to be in flight: the caller's next
// counted s_waitcnt names it ("+v") before anything reads it
__device__ __forceinline__ void gload16(u32x4& r, uint32_t voff, const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(base) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_u(uint32_t addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// six cross terms of the exact product, smallest first: term i multiplies weight plane TP[i] with activation plane TQ[i]
constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};
// slot (within a column block of 6 MT slots) after whose matrix instruction plane p of that block's weight fragment is dead
constexpr int last_use(int plane, int mt) { return (plane == 0 ? 5 : plane == 1 ? 3 : 0) * mt + mt - 1; }
template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void for_each_ic(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }

// The kernel body, for workgroup `bx` of `Gx` (one per CU; Gx a multiple of 8 so that bx & 7 is the XCD) and reduce split z:
// shared by the one-GEMM kernel below and by the chain kernel (gemm_bf16_chain.hpp), which walks a LIST of such problems.
// g.tiles_n = N / BN column blocks; the reduce range of a split is a multiple of 256, >= 512.
// Every fixed register the body names (a0..a223) is dead on entry and on return: the `panel_state_dead` markers tell
// tools/check_wide_hazards.py so (a caller may run compiler-allocated code -- the chain kernel's attention stages -- between two calls).
template <int MT, int NT, int NS, class Epi>
__device__ __forceinline__ void panel_body(const GemmShape& g, const Epi& epi, const int bx, const int Gx, const int z,
                                           unsigned char* const smem_raw) {
  static_assert(NS == 3, "exact mode only (so far)");
  using G = Geo<MT, NT, NS>;
  constexpr int BM = G::BM, BN = G::BN, SLOT = G::SLOT, WAVE_LDS = G::WAVE_LDS, DREQ = G::DREQ, NTERM = G::NTERM;
  constexpr int NSLOT = G::NSLOT, NB = G::NB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  asm volatile("s_nop 0 ; panel_state_dead" ::: "memory");

  int Meff = g.M;
  if (g.count) Meff = min(Meff, *g.count);
  const int tiles_m_live = (Meff + BM - 1) / BM;
  const int T = tiles_m_live * g.tiles_n;                  // live tiles, row panel fastest: t = tile_n * tiles_m_live + tile_m
  // workgroup b owns the tiles [T b' / G, T (b' + 1) / G) with b' = the XCD-major index of b: an XCD's workgroups walk a
  // contiguous eighth of the list, i.e. share a few column blocks of the weights (which every panel re-reads) in their L2
  const int bq = ((Gx & 7) == 0) ? (bx & 7) * (Gx >> 3) + (bx >> 3) : bx;
  const int t_beg = (int)((int64_t)bq * T / Gx), t_end = (int)((int64_t)(bq + 1) * T / Gx);
  const int kbeg = z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int C = __builtin_amdgcn_readfirstlane((kend - kbeg) >> 8);        // 64-deep chunks per wave and tile (the launcher: >= 2, whole)

  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const uint32_t ring = lds0 + wave * WAVE_LDS;

  // dead rows (a live-row list lists them behind the live ones): zero-filled, every WAVE of the grid its share of the rows --
  // the row indices of a round of four rows are requested together (a load -> store chain per element took the workgroup
  // longer than a tile), a row's float4s go to the lanes
  auto zero_dead_rows = [&]() {
    if (g.zero_dead && g.rows && z == 0) {
      const int n_dead = g.M - Meff, n4 = (g.N + 3) / 4;
      const int nw = Gx * 4, w0 = bx * 4 + wave;
      for (int j = w0; j < n_dead; j += 4 * nw) {
        int ph[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ph[u] = g.rows[Meff + min(j + u * nw, n_dead - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u * nw < n_dead)
            for (int c4 = lane; c4 < n4; c4 += 64) epi.zero4(ph[u], 4 * c4, g.N);
      }
    }
  };
  if (t_beg >= t_end) { zero_dead_rows(); return; }
  {
    // ---- weight fragments: byte address = Pfrag + k16 * ldpf * 2 + ((tile_n * NT + nt) * NS + plane) * 1024 + lane * 16
    uint32_t bvoff[(NB + 3) / 4];
#pragma unroll
    for (int j = 0; j < (NB + 3) / 4; ++j) bvoff[j] = lane * 16 + j * 4096;
    const int64_t ldpf_b = g.ldpf * 2;
    const char* const pf0 = reinterpret_cast<const char*>(g.Pfrag) + (int64_t)((kbeg >> 4) + 4 * wave) * ldpf_b;
    const char* const q_src = reinterpret_cast<const char*>(g.Q + kbeg) + wave * 256;       // + c * 1024 for chunk c
    const int* const rows_base = g.rows ? g.rows : reinterpret_cast<const int*>(g.Q);        // (no list: the load's result is not used)
    const bool has_rows = g.rows != nullptr;
    const uint32_t ldq4 = (uint32_t)(g.ldq * 4);

    // current / next tile (the last tile's "next" is itself: its prefetches are harmless repeats)
    int t_cur = t_beg;
    int tn_c = t_cur / tiles_m_live, tm_c = t_cur - tn_c * tiles_m_live;
    int t_nx = min(t_cur + 1, t_end - 1);
    int tn_n = t_nx / tiles_m_live, tm_n = t_nx - tn_n * tiles_m_live;
    const char* pf_cur = pf0 + (int64_t)tn_c * (NT * NS * 1024);      // + (16 c + i) * ldpf_b for step i of chunk c
    const char* pf_nxt = pf0 + (int64_t)tn_n * (NT * NS * 1024);

    // ---- live rows, one per lane (row lane & (BM - 1) of the panel): physical row, and its byte offset in Qop
    auto row_index = [&](int tm) { return min(tm * BM + (lane & (BM - 1)), Meff - 1); };
    int ph_cur = has_rows ? g.rows[row_index(tm_c)] : row_index(tm_c);
    uint32_t offc = (uint32_t)ph_cur * ldq4;
    int ph_nxt = ph_cur;
    uint32_t offn = offc;
    uint32_t rvoff = (uint32_t)row_index(tm_n) * 4;           // the next tile's row-list entry of this lane
    int ph_tmp = 0;

    // the weights of steps 0 and 1: nothing they depend on
    for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int IDX = decltype(i_c)::value; bload<IDX, (IDX & 3) * 1024>(bvoff[IDX >> 2], pf_cur); },
                std::make_integer_sequence<int, NB>{});
    for_each_ic([&](auto i_c) __attribute__((always_inline)) { constexpr int IDX = decltype(i_c)::value; bload<NB + IDX, (IDX & 3) * 1024>(bvoff[IDX >> 2], pf_cur + ldpf_b); },
                std::make_integer_sequence<int, NB>{});

    // ---- request q of a chunk covers rows 4 q .. 4 q + 3 (256 B each): lane -> (row 4 q + (lane >> 4), 16-byte slot lane & 15);
    //      the row's offset comes from the lane that holds it (ds_bpermute); the LDS image swizzles the slot with row & 15
    const int bp_idx = 4 * (lane >> 4);                        // + 16 q: byte index of lane 4 q + (lane >> 4)
    uint32_t wdst[4];                                          // LDS byte offset of this lane's slot in rows 4 q + (lane >> 4), by q & 3 (+ q * 1024)
#pragma unroll
    for (int j = 0; j < 4; ++j) wdst[j] = (lane >> 4) * 256 + 16 * ((lane & 15) ^ ((4 * j + (lane >> 4)) & 15));
    uint32_t doff[DREQ];
    u32x4 stg[DREQ];                                           // a chunk on its way: requested in step 3, stored to the ring in step 1
#pragma unroll
    for (int q = 0; q < DREQ; ++q) doff[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_idx + 16 * q, (int)offc) + 16 * (lane & 15);
    // chunks 0 and 1 of the first tile, requested together; `stg` ends up holding chunk 0: the first trip's step 1 stores it
    // "again" (the stream's chunk c + 2 of a trip before the first) to where it already is
    u32x4 stg1[DREQ];
#pragma unroll
    for (int q = 0; q < DREQ; ++q) gload16(stg[q], doff[q], q_src);
#pragma unroll
    for (int q = 0; q < DREQ; ++q) gload16(stg1[q], doff[q], q_src + 1024);
    zero_dead_rows();                 // (stores only: they ride under the first loads' latency)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for_each_ic([&](auto q_c) __attribute__((always_inline)) { constexpr int Qi = decltype(q_c)::value;
                  asm volatile("" : "+v"(stg[Qi]), "+v"(stg1[Qi]));
                  lds_write_u<Qi * 1024>(ring + wdst[Qi & 3], stg[Qi]);
                  lds_write_u<SLOT + Qi * 1024>(ring + wdst[Qi & 3], stg1[Qi]); }, std::make_integer_sequence<int, DREQ>{});

    // ---- raw fragment addresses: row l31 (+ 32 mt: immediate), k-step i, half j: slot (4 i + 2 lh + j) ^ (l31 & 15)
    uint32_t arow[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) arow[i][j] = ring + l31 * 256 + 16 * (((4 * i + 2 * lh + j) ^ (l31 & 15)) & 15);
    int toggle = SLOT;          // arow += toggle moves the addresses to the other chunk of the ring
    uint32_t slot_cur = ring;   // the chunk the current trip reads
    uint32_t slot_fill = ring;  // the chunk requested in the previous trip's step 3 (stored in this trip's step 1)

    u32x4 qa[MT][3], qb[MT][3];     // split activation fragments [row block][term]: one set in use, the other being built
    f32x4 raw[MT][2];               // raw fp32 fragment of a row block: [half]
    float tmp[4];
    uint32_t bpv[DREQ];

    // vector operation `op` (0..43) of the exact three-way split of row block u into q[u][.] (gemm_bf16_wide.hpp: split_op)
    auto split_op = [&](u32x4 (&q)[MT][3], int u, int op) __attribute__((always_inline)) {
      auto pack = [&](int plane, int pr) {
        q[u][plane][pr] = bf16_pack_top(raw[u][pr >> 1][(pr & 1) * 2], raw[u][pr >> 1][(pr & 1) * 2 + 1]);
      };
      if (op < 4) pack(0, op);
      else if (op < 20 || (op >= 24 && op < 40)) {
        const int o = (op < 20) ? op - 4 : op - 24;       // [element 0 | 1][and | sub][pair]
        const int el = o >> 3, sub = (o >> 2) & 1, pr = o & 3;
        const float x = raw[u][pr >> 1][(pr & 1) * 2 + el];
        if (!sub) tmp[pr] = f32_top16(x);
        else raw[u][pr >> 1][(pr & 1) * 2 + el] = x - tmp[pr];
      }
      else if (op < 24) pack(1, op - 20);
      else pack(2, op - 40);
    };

    // ---- one k-step (I = 0..3 of chunk c): NSLOT matrix-instruction slots = NT column blocks x 6 terms x MT row blocks.
    //      In the shadow of the matrix instructions, all of it for LATER steps of the stream (which runs on into the next tile):
    //        the weight fragments of step t + 2, each into the registers of the fragment that was just used last
    //        the raw activation fragments of step t + 1 (row block u: window u of NSLOT / MT slots), their split
    //        I == 0: the next tile's row-list entry of this lane is requested
    //        I == 2: it has arrived; the row offsets of chunk c + 2 go to the lanes that will request them (ds_bpermute);
    //                the raw-fragment addresses move to the other chunk of the ring
    //        I == 3: the DMA requests of chunk c + 2 into the chunk that step (c, 2) read last
    //      ONE wait for memory, at the start: everything but what the PREVIOUS step issued has arrived (the weights of this
    //      step are two steps old; the chunk a step I == 3 starts to read is older still)
    constexpr int WL = NSLOT / MT, RW = WL >= 24 ? 6 : 4, PER = (44 + (WL - RW) - 1) / (WL - RW);
    constexpr int BP0 = NSLOT / 3, BP1 = 2 * NSLOT / 3;          // I == 2: ds_bpermute in slots [BP0, BP0 + DREQ), their use from BP1 on
    // everything run-time a trip needs is selected BEFORE its four steps (bnext[], dma_src, off_sel): a branch inside the
    // trip splits it into basic blocks, and hipcc then sinks the split operations of one step out of the matrix
    // instructions' shadow, down to the block that uses them (400 cycles per step, measured)
    const char* bnext[4];
    bool chunk_next_t = false;
    const char* dma_src = q_src;
    uint32_t off_sel = offc;
    auto step = [&](auto i_c, auto first_c) __attribute__((always_inline)) {
      constexpr int I = decltype(i_c)::value;
      constexpr bool FIRST = decltype(first_c)::value != 0 && I == 0;       // the tile's first step: term 0 starts the accumulators
      constexpr int ST = I & 1;
      constexpr int NBe = (PANEL_ABL & 8) ? 0 : NB;
      constexpr int n_prev = I == 0 ? NBe + ((PANEL_ABL & 1) ? 0 : DREQ) : (I == 1 ? NBe + ((PANEL_ABL & 2) ? 0 : 1) : NBe);
      auto& QC = ST ? qb : qa;
      auto& QN = ST ? qa : qb;
      const char* const b_next = bnext[I];
      auto slot = [&](auto gs_c) __attribute__((always_inline)) {
        constexpr int gs = decltype(gs_c)::value;
        constexpr int nt = gs / (NTERM * MT), sl = gs % (NTERM * MT), term = sl / MT, mt = sl % MT;
        if constexpr (gs == 0) {
          if constexpr (I == 2) asm volatile("s_waitcnt vmcnt(%1)\n\ts_nop 1" : "+v"(ph_tmp) : "n"(n_prev) : "memory");
          else if constexpr (I == 1 && DREQ == 16)
            asm volatile("s_waitcnt vmcnt(%16)\n\ts_nop 1" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]),
                         "+v"(stg[8 % DREQ]), "+v"(stg[9 % DREQ]), "+v"(stg[10 % DREQ]), "+v"(stg[11 % DREQ]), "+v"(stg[12 % DREQ]), "+v"(stg[13 % DREQ]), "+v"(stg[14 % DREQ]), "+v"(stg[15 % DREQ])
                         : "n"(n_prev) : "memory");
          else if constexpr (I == 1)
            asm volatile("s_waitcnt vmcnt(%8)\n\ts_nop 1" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7])
                         : "n"(n_prev) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)\n\ts_nop 1" : : "n"(n_prev) : "memory");
        }
        // the raw fragments of row block u have arrived (the registers ride along so that no use moves above)
        if constexpr (gs == RW) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]) :: "memory");
        if constexpr (MT == 2 && gs == WL + RW) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[MT - 1][0]), "+v"(raw[MT - 1][1]) :: "memory");
        if constexpr (FIRST && term == 0) pmfma0<mt * NT + nt, ST * NB + nt * NS + TP[term]>(QC[mt][TQ[term]]);
        else pmfma<mt * NT + nt, ST * NB + nt * NS + TP[term]>(QC[mt][TQ[term]]);
        {
          constexpr int U = gs / WL, ws = gs % WL;          // row block whose window this slot is in
          if constexpr (ws == 0) lds_read<U * 32 * 256>(raw[U][0], arow[(I + 1) & 3][0]);
          if constexpr (ws == 1) lds_read<U * 32 * 256>(raw[U][1], arow[(I + 1) & 3][1]);
          if constexpr (ws >= RW && !(PANEL_ABL & 4)) {
            constexpr int first = (ws - RW) * PER;
            if constexpr (first + 0 < 44 && PER > 0) split_op(QN, U, first + 0);
            if constexpr (first + 1 < 44 && PER > 1) split_op(QN, U, first + 1);
            if constexpr (first + 2 < 44 && PER > 2) split_op(QN, U, first + 2);
            if constexpr (first + 3 < 44 && PER > 3) split_op(QN, U, first + 3);
            if constexpr (first + 4 < 44 && PER > 4) split_op(QN, U, first + 4);
            if constexpr (first + 5 < 44 && PER > 5) split_op(QN, U, first + 5);
            static_assert(PER <= 6, "split operations per slot");
          }
        }
        if constexpr (!(PANEL_ABL & 8)) {
        if constexpr (sl == last_use(0, MT)) bload<ST * NB + nt * NS + 0, ((nt * NS + 0) & 3) * 1024>(bvoff[(nt * NS + 0) >> 2], b_next);
        if constexpr (sl == last_use(1, MT)) bload<ST * NB + nt * NS + 1, ((nt * NS + 1) & 3) * 1024>(bvoff[(nt * NS + 1) >> 2], b_next);
        if constexpr (sl == last_use(2, MT)) bload<ST * NB + nt * NS + 2, ((nt * NS + 2) & 3) * 1024>(bvoff[(nt * NS + 2) >> 2], b_next);
        }
        if constexpr (I == 0 && gs == 2 && !(PANEL_ABL & 2)) asm volatile("global_load_dword %0, %1, %2" : "=v"(ph_tmp) : "v"(rvoff), "s"(rows_base) : "memory");
        if constexpr (I == 2 && !(PANEL_ABL & 2)) {
          if constexpr (gs == 1) {
            ph_nxt = has_rows ? ph_tmp : (int)(rvoff >> 2);
            offn = (uint32_t)ph_nxt * ldq4;
            off_sel = chunk_next_t ? offn : offc;
          }
          if constexpr (gs >= BP0 && gs < BP0 + DREQ) bpv[gs - BP0] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_idx + 16 * (gs - BP0), (int)off_sel);
          if constexpr (gs >= BP1 && gs < BP1 + DREQ) doff[gs - BP1] = bpv[gs - BP1] + 16 * (lane & 15);
        }
        if constexpr (I == 2) {
          if constexpr (gs >= NSLOT - 8) { constexpr int A = gs - (NSLOT - 8); arow[A >> 1][A & 1] += toggle; }
        }
        if constexpr (I == 3 && !(PANEL_ABL & 1)) {
          // request d of chunk c + 2 goes out at slot 3 d ...
          if constexpr (gs % 3 == 0 && gs / 3 < DREQ) gload16(stg[gs / 3], doff[gs / 3], dma_src);
        }
        if constexpr (I == 1 && !(PANEL_ABL & 1)) {
          // ... and is stored, two steps later, into the chunk that step (c - 1, 2) read last
          if constexpr (gs % 3 == 1 && gs / 3 < DREQ) lds_write_u<(gs / 3) * 1024>(slot_fill + wdst[(gs / 3) & 3], stg[gs / 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for_each_ic(slot, std::make_integer_sequence<int, NSLOT>{});
      if constexpr (I == 2) toggle = -toggle;
    };

    // ---- split step 0 with nothing to hide behind
    lds_read<0>(raw[0][0], arow[0][0]);
    lds_read<0>(raw[0][1], arow[0][1]);
    if constexpr (MT == 2) {
      lds_read<32 * 256>(raw[MT - 1][0], arow[0][0]);
      lds_read<32 * 256>(raw[MT - 1][1], arow[0][1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[MT - 1][0]), "+v"(raw[MT - 1][1]) :: "memory");
#pragma unroll
    for (int op = 0; op < 44; ++op) split_op(qa, 0, op);
    if constexpr (MT == 2) {
#pragma unroll
      for (int op = 0; op < 44; ++op) split_op(qa, MT - 1, op);
    }

#if defined(PANEL_TRACE) && PANEL_TRACE >= 2
    unsigned long long step_cyc[4] = {0, 0, 0, 0};
#endif
#pragma nounroll
    for (;;) {
      PANEL_MARK(t_cur, 0);
      auto trip = [&](auto first_c, const int c) __attribute__((always_inline)) {
        {
          const bool into_next = c + 1 >= C;                            // steps t + 2 of I >= 2 belong to the next tile
          chunk_next_t = c + 2 >= C;                                    // chunk c + 2 belongs to the next tile
          const char* const cur = pf_cur + (int64_t)(16 * c) * ldpf_b;
          bnext[0] = cur + 2 * ldpf_b;
          bnext[1] = cur + 3 * ldpf_b;
          const char* const nx = into_next ? pf_nxt : cur + 16 * ldpf_b;
          bnext[2] = nx;
          bnext[3] = nx + ldpf_b;
          dma_src = q_src + (int64_t)(chunk_next_t ? c + 2 - C : c + 2) * 1024;
        }
        PANEL_STEP_T(s0);
        step(IC<0>{}, first_c);
        PANEL_STEP_T(s1);
        step(IC<1>{}, first_c);
        PANEL_STEP_T(s2);
        step(IC<2>{}, first_c);
        PANEL_STEP_T(s3);
        step(IC<3>{}, first_c);
        PANEL_STEP_T(s4);
        PANEL_STEP_ACC(0, s0, s1); PANEL_STEP_ACC(1, s1, s2); PANEL_STEP_ACC(2, s2, s3); PANEL_STEP_ACC(3, s3, s4);
        slot_fill = slot_cur;
        slot_cur = slot_cur == ring ? ring + SLOT : ring;
      };
      trip(IC<1>{}, 0);
#pragma nounroll
      for (int c = 1; c < C; ++c) trip(IC<0>{}, c);
      PANEL_MARK(t_cur, 2);
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");     // the last matrix instruction has written its rows

      // ---- epilogue, half the column blocks (all row blocks) per pass: the four waves' partial sums -> LDS, straight from the
      //      accumulation file, into the chunk of each wave's ring that is free now (`slot_fill`: its next contents are in
      //      `stg` until the next tile's step 1) as [BM][64 fp32], 16-byte slots swizzled with row & 7; barrier; every thread
      //      adds the four partials of its float4s in wave order; barrier; fused epilogue (its stores overlap the next pass).
      //      Accumulator map (a = weight fragment, b = activation fragment): register r of acc[mt][nt] is row mt*32 + l31,
      //      column nt*32 + 8 (r >> 2) + 4 lh + (r & 3)
      {
        constexpr int NPASS = NT / 2;                 // 64 columns per pass
        constexpr int TM = BM / 16;                   // float4s per thread and pass: rows r0 + 16 a, one column group
        const int m0 = tm_c * BM, n0 = tn_c * BN;
        const uint32_t pw = slot_fill + l31 * 256;
        const int r0 = tid >> 4, ch = tid & 15;
        const unsigned char* const pr = smem_raw + (slot_fill - ring) + r0 * 256 + 16 * (ch ^ (r0 & 7));      // + wave * WAVE_LDS (rows r0 + 16 a: same swizzle)
        int mlog[TM], mphys[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          mlog[a] = m0 + r0 + 16 * a;
          mphys[a] = __builtin_amdgcn_ds_bpermute(4 * (r0 + 16 * a), ph_cur);
        }
        const bool fast = epi.fast_ok() && m0 + BM <= Meff && n0 + BN <= g.N;
        for_each_ic([&](auto p_c) __attribute__((always_inline)) {
          constexpr int P = decltype(p_c)::value;
          for_each_ic([&](auto i_c) __attribute__((always_inline)) {
            constexpr int A = decltype(i_c)::value / 8, B2 = (decltype(i_c)::value / 4) & 1, QD = decltype(i_c)::value & 3;      // row block, column block of the pass, quad
            acc_store<A * NT + 2 * P + B2, QD, A * 32 * 256>(pw + 16 * ((8 * B2 + 2 * QD + lh) ^ (l31 & 7)));
          }, std::make_integer_sequence<int, 8 * MT>{});
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __syncthreads();
          f32x4 acc4[TM][1];
#pragma unroll
          for (int a = 0; a < TM; ++a) {
            f32x4 s = *reinterpret_cast<const f32x4*>(pr + a * 16 * 256);
#pragma unroll
            for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(pr + a * 16 * 256 + w * WAVE_LDS);
            acc4[a][0] = s;
          }
          __syncthreads();
          int ncol[1] = {n0 + 64 * P + 4 * ch};
          if (fast) epi.template tile_fast<TM, 1, true>(acc4, mphys, ncol, g.N, z);
          else epilogue_all<0, TM, 1, true, Epi>(epi, acc4, mlog, mphys, ncol, Meff, g.N, z);
        }, std::make_integer_sequence<int, NPASS>{});
      }
      PANEL_MARK(t_cur, 3);
#if defined(PANEL_TRACE) && PANEL_TRACE >= 2
      if (g_panel_trace && tid == 0 && t_cur + 1 >= t_end) for (int i = 0; i < 4; ++i) g_panel_trace[(size_t)(8192 + bx) * 8 + i] = step_cyc[i];
#endif
      if (t_cur + 1 >= t_end) break;
      // ---- the next tile becomes the current one
      t_cur = t_nx; tn_c = tn_n; tm_c = tm_n; pf_cur = pf_nxt; ph_cur = ph_nxt; offc = offn;
      t_nx = min(t_cur + 1, t_end - 1);
      tn_n = t_nx / tiles_m_live; tm_n = t_nx - tn_n * tiles_m_live;
      pf_nxt = pf0 + (int64_t)tn_n * (NT * NS * 1024);
      rvoff = (uint32_t)row_index(tm_n) * 4;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the repeats requested behind the last tile have landed
  }
  asm volatile("s_nop 0 ; panel_state_dead" ::: "memory");
}

// one GEMM per launch: gridDim.x workgroups (one per CU), gridDim.z reduce splits
template <int MT, int NT, int NS, class Epi>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(GemmShape g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  panel_body<MT, NT, NS, Epi>(g, epi, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z, smem_raw);
}
}  // namespace panel
