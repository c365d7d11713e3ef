// Weight-gradient GEMM body on fp32 operands: C[i][j] = sum over the live rows r of A[r][i] * B[r][j]  (dW = dZ^T X).
// The eight-phase schedule of gemm_g256.hpp (256 x 256 workgroup tile, both operands global -> LDS by DMA, counted vmcnt, raw
// s_barrier, the two wave rows one barrier apart) with
//   * fp32 tiles in LDS: a k-tile is 32 reduce rows, a half-tile 32 rows x 128 columns = 16 KiB (512-byte LDS rows; the 16-byte
//     chunks of a row XOR-swizzled by ((r >> 3) & 1) << 2 on the DMA's source address, so that the two 16-lane halves of a
//     32-lane read group -- rows r and r + 8 -- hit different bank halves);
//   * the LIVE-ROW GATHER inside the DMA: every request carries its own row offset (scalar loads of the row list, rows past the
//     live count are requested out of bounds and arrive as zeros) -- no packed copy of the operands, no dead-row work;
//   * fragments built in registers: a lane reads its 8 reduce rows of one column (ds_read2st64_b32), splits them into NS bf16
//     planes (NS = 1: round to nearest even, the throughput mode; NS = 3: the exact truncation split x = h + m + l of
//     gemm_bf16.hpp, six cross products per block, smallest first) and feeds v_mfma_f32_16x16x32_bf16: the reduce index is the
//     ROW of both operands, so this register transpose replaces the transposing stager of the 128 x 128 kernel;
//   * dZ's column sums (the bias gradient) in the first column tile, straight from the fp32 LDS image: thread t adds 16 rows of
//     column t & 255 per k-tile in phase 2 (one accumulator; the light phase of the schedule).
//   Per k-tile a wave issues 32 x NS(NS+1)/2 matrix instructions (NS = 3: 192 = 3072 cycles) against 48 LDS read instructions and
//   ~530 vector instructions of splitting, which run while the SIMD's other wave is in its matrix block.
#pragma once
#include "gemm_g256.hpp"
#include "gemm_bf16.hpp"

namespace g256w {

using g256::f32x4;
using g256::lds_ptr;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 32, THREADS = 512;
constexpr int HALF = 32 * 512;             // bytes of a half-tile: 32 rows x 128 fp32
constexpr int BUF = 4 * HALF;              // one k-tile: A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF + 2048;      // + the 512 column-sum accumulators of the bias gradient

struct Operand {
  __amdgpu_buffer_rsrc_t rsrc;             // base = the tile's first column, extent = the whole matrix
  unsigned colpart;                        // per lane: byte offset of its 16-byte chunk inside a 512-byte half-row
  unsigned rowstep;                        // ld * 4
};
__device__ __forceinline__ Operand operand(const float* p, int64_t ld, int col0, int rows_total, int lane, int wave) {
  Operand o;
  const size_t all = (size_t)rows_total * (size_t)ld * 4, off = (size_t)col0 * 4;
  const size_t bytes = all > off ? all - off : 0;
  // the descriptor's inputs through readfirstlane: values the compiler computed on the vector unit would otherwise wrap every
  // DMA request in a waterfall loop (cdna_hip_programming.md, T20)
  o.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g256::uniform_ptr(p + col0), 0,
                                             (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xfffffff0u ? bytes : 0xfffffff0u)), 0x00020000);
  o.colpart = (unsigned)(((lane & 31) ^ (((wave >> 1) & 1) << 2)) << 4);
  o.rowstep = (unsigned)ld * 4u;
  return o;
}

// the two physical rows (request j = 0, 1) a lane fetches for reduce tile `tile`; -1: past the live rows (fetched as zeros)
struct Rows { int r[2]; };
// the live-row list behind SCALAR loads: the CONSTANT address space (nothing writes the list while the kernel runs) makes the
// compiler pick s_load_dword for these wave-uniform addresses -- a vector load here would make it drain the DMA queue at every use
typedef const __attribute__((address_space(4))) int* const_int_ptr;
struct RowList { const_int_ptr p; int cap; };
__device__ __forceinline__ RowList row_list(const int* list, int rows_cap) {
  RowList l;
  l.p = (const_int_ptr)(uintptr_t)g256::uniform_ptr(list);
  l.cap = rows_cap;
  return l;
}
__device__ __forceinline__ Rows rows_of_tile(int tile, int n_live, const RowList& list, int lane, int wave) {
  Rows o;
  const int base = tile * BK + wave * 4;      // wave-uniform
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r0 = base + 2 * j, r1 = r0 + 1;
    int p0 = r0, p1 = r1;
    if (list.p) {                              // (entries past the live count are masked below)
      p0 = list.p[min(r0, list.cap - 1)];
      p1 = list.p[min(r1, list.cap - 1)];
    }
    const int r = (lane & 32) ? r1 : r0;
    o.r[j] = r < n_live ? ((lane & 32) ? p1 : p0) : -1;
  }
  return o;
}

template <int BUFI, int OPND, int H>
__device__ __forceinline__ void stage(unsigned char* smem, const Operand& op, unsigned lds_stage, const Rows& rw) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned voff = rw.r[j] >= 0 ? __umul24((unsigned)rw.r[j], op.rowstep) + op.colpart : 0x80000000u;      // (rows, row pitch < 2^24)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(op.rsrc, (lds_ptr)(smem + BUFI * BUF + OPND * 2 * HALF + H * HALF + lds_stage + j * 1024), 16, voff, H * 512, 0,
                                             0);
  }
}

// one 16-column x 32-row fragment: the lane's 8 rows of its column, split into NS planes of 8 bf16 (4 dwords each)
template <int NS>
__device__ __forceinline__ void read_frag(const unsigned char* addr, u32x4_t (&pl)[NS]) {
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = *reinterpret_cast<const float*>(addr + i * 512);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t w[NS];
    bf16_split2<NS>(e[2 * q], e[2 * q + 1], w);
#pragma unroll
    for (int p = 0; p < NS; ++p) pl[p][q] = w[p];
  }
}

struct Frag { unsigned a_even, a_odd, b_even, b_odd; };      // per-lane LDS addresses inside a half-tile
__device__ __forceinline__ Frag frag_addr(int lane, int wr, int wc) {
  Frag f;
  const int g = lane >> 4, c = lane & 15, x = g & 1;
  const unsigned row = (unsigned)(g * 8) * 512u;
  // dword index inside the 128-column half-row: bits 0-3 lane column, bits 4-5 fragment (mf | nf), bit 6 (A) / bits 5-6 (B) the wave;
  // the swizzle flips bit 4 for the odd row groups
  f.a_even = row + (unsigned)(((wr * 64 + c) ^ (x << 4)) << 2);
  f.a_odd = row + (unsigned)(((wr * 64 + 16 + c) ^ (x << 4)) << 2);
  f.b_even = row + (unsigned)(((wc * 32 + c) ^ (x << 4)) << 2);
  f.b_odd = row + (unsigned)(((wc * 32 + 16 + c) ^ (x << 4)) << 2);
  return f;
}

template <int NS, int BUFI, int H>
__device__ __forceinline__ void read_a(const unsigned char* smem, const Frag& f, u32x4_t (&a)[4][NS]) {
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) read_frag<NS>(smem + BUFI * BUF + H * HALF + ((mf & 1) ? f.a_odd : f.a_even) + (mf >> 1) * 128, a[mf]);
}
template <int NS, int BUFI, int G>
__device__ __forceinline__ void read_b(const unsigned char* smem, const Frag& f, u32x4_t (&b)[2][NS]) {
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) read_frag<NS>(smem + BUFI * BUF + 2 * HALF + G * HALF + (nf ? f.b_odd : f.b_even), b[nf]);
}
// the bias gradient: thread t adds rows (t >> 8) * 16 .. + 15 of column t & 255 of the A k-tile, straight from the fp32 image
// (phase 2: both halves of A are in place, nothing restages them before phase 3), into ITS accumulator in LDS: the exact-mode
// kernel has no register left to carry a sum across the k-loop
template <int BUFI>
__device__ __forceinline__ void colsum_rows(unsigned char* smem, int wave) {
  // (the thread index rebuilt from v_mbcnt + the wave number: kept across the loop it is spilled in the exact-mode kernel, and a
  //  spill reload costs a vmcnt(0), i.e. the DMA pipeline)
  const int tid = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int c = tid & 255, cc = c & 127;
  const unsigned off = (unsigned)(BUFI * BUF + (c >> 7) * HALF + (tid >> 8) * (16 * 512) + (cc << 2));
  const unsigned char* p0 = smem + off;                     // rows 0..7 of the 16: (r >> 3) & 1 = 0
  const unsigned char* p1 = smem + (off ^ 64u);             // rows 8..15: column ^ 16
  float* accp = reinterpret_cast<float*>(smem + 2 * BUF) + tid;
  float s = *accp;      // one chain of adds (two interleaved chains get packed into v_pk_add_f32 with register shuffles around them)
#pragma unroll
  for (int r = 0; r < 8; ++r) s += *reinterpret_cast<const float*>(p0 + r * 512);
#pragma unroll
  for (int r = 8; r < 16; ++r) s += *reinterpret_cast<const float*>(p1 + r * 512);
  *accp = s;
}

#define G256W_MFMA(PB, PA)                                                                                                                        \
  _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) acc[MH * 4 + mf][NH * 2 + nf] =              \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g256::bf16x8, b[nf][PB]), __builtin_bit_cast(g256::bf16x8, a[mf][PA]),           \
                                              acc[MH * 4 + mf][NH * 2 + nf], 0, 0, 0)
template <int NS, int MH, int NH>
__device__ __forceinline__ void quadrant(f32x4 (&acc)[8][4], const u32x4_t (&a)[4][NS], const u32x4_t (&b)[2][NS]) {
  __builtin_amdgcn_s_setprio(1);
  if constexpr (NS == 1) {
    G256W_MFMA(0, 0);
  } else {      // six cross terms, smallest first (gemm_bf16.hpp: compute)
    G256W_MFMA(2, 0);
    G256W_MFMA(0, 2);
    G256W_MFMA(1, 1);
    G256W_MFMA(1, 0);
    G256W_MFMA(0, 1);
    G256W_MFMA(0, 0);
  }
  __builtin_amdgcn_s_setprio(0);
}

#define G256W_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// the barrier between a phase's load part and its matrix block: the LDS reads are retired (what they read may be restaged by the
// other wave row in ITS next phase) and the splitting stays in front of it (it runs beside the other wave row's matrix block)
#define G256W_BARRIER_A() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// What a workgroup reduces over: its k-tiles are tile0, tile0 + step, ... (nk of them) of the live rows
struct Walk { int tile0, step, nk, n_live; RowList list; };

template <int NS, bool SUM, int BUFI>
__device__ __forceinline__ void ktile(unsigned char* smem, const Operand& oa, const Operand& ob, const Frag& f, unsigned lds_stage, const Walk& w, int kt,
                                      Rows& rw, f32x4 (&acc)[8][4], u32x4_t (&a)[4][NS], u32x4_t (&b0)[2][NS], u32x4_t (&b1)[2][NS], int lane, int wave) {
  const bool next1 = kt + 1 < w.nk, next2 = kt + 2 < w.nk;      // wave-uniform
  // ---- phase 1   (rw = the rows of tile kt + 1 on entry)
  read_b<NS, BUFI, 0>(smem, f, b0);
  if constexpr (NS == 3) __builtin_amdgcn_sched_barrier(0);      // B0's raw fp32 registers are dead before A0's are loaded (register budget)
  read_a<NS, BUFI, 0>(smem, f, a);
  if (next1) stage<BUFI ^ 1, 0, 1>(smem, oa, lds_stage, rw);
  if (next2) rw = rows_of_tile(w.tile0 + (kt + 2) * w.step, w.n_live, w.list, lane, wave);
  G256W_BARRIER_A();
  quadrant<NS, 0, 0>(acc, a, b0);
  G256W_BARRIER();
  // ---- phase 2
  read_b<NS, BUFI, 1>(smem, f, b1);
  if constexpr (SUM) {
    __builtin_amdgcn_sched_barrier(0);      // (B1's raw registers are free again)
    colsum_rows<BUFI>(smem, wave);
  }
  if (next2) stage<BUFI, 1, 0>(smem, ob, lds_stage, rw);
  G256W_BARRIER_A();
  quadrant<NS, 0, 1>(acc, a, b1);
  G256W_BARRIER();
  // ---- phase 3
  read_a<NS, BUFI, 1>(smem, f, a);
  if (next2) stage<BUFI, 0, 0>(smem, oa, lds_stage, rw);
  G256W_BARRIER_A();
  quadrant<NS, 1, 1>(acc, a, b1);
  G256W_BARRIER();
  // ---- phase 4
  if (next2) stage<BUFI, 1, 1>(smem, ob, lds_stage, rw);
  if (next2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G256W_BARRIER();
  quadrant<NS, 1, 0>(acc, a, b0);
  G256W_BARRIER();
}

// The same k-tile in TWO phases (two quadrants = 32 x NS(NS+1)/2 matrix instructions per barrier pair): four barriers per k-tile
// instead of eight.  Staging moves to two half-tiles per phase: phase A(t) stages {A0, A1}[t+1] into the other buffer (A0's last
// read was phase A(t-1), A1's phase B(t-1): retired before that phase's first barrier), phase B(t) stages {B0, B1}[t+2] into its
// own (read in phase A(t)).  Counted waits: phase A leaves {B[t+1], A[t+1]} = 8 requests in flight (A1[t] is retired: read in
// phase B), phase B leaves {A1[t+1], B[t+2]} = 6 (B[t+1] and A0[t+1] are retired: read in phase A(t+1)).
template <int NS, bool SUM, int BUFI>
__device__ __forceinline__ void ktile2(unsigned char* smem, const Operand& oa, const Operand& ob, const Frag& f, unsigned lds_stage, const Walk& w, int kt,
                                       Rows& rw, f32x4 (&acc)[8][4], u32x4_t (&a)[4][NS], u32x4_t (&b0)[2][NS], u32x4_t (&b1)[2][NS], int lane, int wave) {
  const bool next1 = kt + 1 < w.nk, next2 = kt + 2 < w.nk;      // wave-uniform
  // ---- phase A   (rw = the rows of tile kt + 1 on entry)
  read_b<NS, BUFI, 0>(smem, f, b0);
  read_b<NS, BUFI, 1>(smem, f, b1);
  if constexpr (NS == 3) __builtin_amdgcn_sched_barrier(0);      // (the raw fp32 registers of B are dead before A0's are loaded)
  read_a<NS, BUFI, 0>(smem, f, a);
  if (next1) {
    stage<BUFI ^ 1, 0, 0>(smem, oa, lds_stage, rw);
    stage<BUFI ^ 1, 0, 1>(smem, oa, lds_stage, rw);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (next2) rw = rows_of_tile(w.tile0 + (kt + 2) * w.step, w.n_live, w.list, lane, wave);
  G256W_BARRIER_A();
  quadrant<NS, 0, 0>(acc, a, b0);
  quadrant<NS, 0, 1>(acc, a, b1);
  G256W_BARRIER();
  // ---- phase B
  read_a<NS, BUFI, 1>(smem, f, a);
  if constexpr (SUM) {
    __builtin_amdgcn_sched_barrier(0);
    colsum_rows<BUFI>(smem, wave);
  }
  if (next2) {
    stage<BUFI, 1, 0>(smem, ob, lds_stage, rw);
    stage<BUFI, 1, 1>(smem, ob, lds_stage, rw);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else if (next1) {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  G256W_BARRIER_A();
  quadrant<NS, 1, 0>(acc, a, b0);
  quadrant<NS, 1, 1>(acc, a, b1);
  G256W_BARRIER();
}

// epi(row, col, v): the four outputs (row, col .. col + 3) of the 256 x 256 tile; episum(row, s) (SUM): the sum of A's column `row` of
// the tile over the workgroup's reduce rows
template <int NS, bool SUM, int PH, class Epi, class EpiSum>
__device__ __forceinline__ void body(unsigned char* smem, const Operand& oa, const Operand& ob, const Walk& w, const Epi& epi, const EpiSum& episum) {
  static_assert(PH == 4 || PH == 2, "phases per k-tile");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const Frag f = frag_addr(lane, wr, wc);
  const unsigned lds_stage = (unsigned)wave * 2048u;      // 4 rows of 512 bytes

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4_t a[4][NS], b0[2][NS], b1[2][NS];
  if constexpr (SUM) reinterpret_cast<float*>(smem + 2 * BUF)[threadIdx.x] = 0.f;      // (ordered before its first use by the prologue barrier)

  // ---- prologue: all of tile 0, then tile 1's B0, A0, B1 (its A1 is staged by tile 0's phase 1)
  Rows rw = rows_of_tile(w.tile0, w.n_live, w.list, lane, wave);
  if constexpr (PH == 4) {
    stage<0, 0, 0>(smem, oa, lds_stage, rw);
    stage<0, 1, 0>(smem, ob, lds_stage, rw);
    stage<0, 1, 1>(smem, ob, lds_stage, rw);
    stage<0, 0, 1>(smem, oa, lds_stage, rw);
    if (w.nk > 1) {
      rw = rows_of_tile(w.tile0 + w.step, w.n_live, w.list, lane, wave);
      stage<1, 1, 0>(smem, ob, lds_stage, rw);
      stage<1, 0, 0>(smem, oa, lds_stage, rw);
      stage<1, 1, 1>(smem, ob, lds_stage, rw);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {      // B[0], A0[0], A1[0], then B[1]: B[0] and A0[0] are retired here, A1[0] by tile 0's phase A
    stage<0, 1, 0>(smem, ob, lds_stage, rw);
    stage<0, 1, 1>(smem, ob, lds_stage, rw);
    stage<0, 0, 0>(smem, oa, lds_stage, rw);
    stage<0, 0, 1>(smem, oa, lds_stage, rw);
    if (w.nk > 1) {
      rw = rows_of_tile(w.tile0 + w.step, w.n_live, w.list, lane, wave);
      stage<1, 1, 0>(smem, ob, lds_stage, rw);
      stage<1, 1, 1>(smem, ob, lds_stage, rw);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  }
  G256W_BARRIER();
  if (wr == 1) G256W_BARRIER();      // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < w.nk; kt += 2) {
    if constexpr (PH == 4) {
      ktile<NS, SUM, 0>(smem, oa, ob, f, lds_stage, w, kt, rw, acc, a, b0, b1, lane, wave);
      if (kt + 1 < w.nk) ktile<NS, SUM, 1>(smem, oa, ob, f, lds_stage, w, kt + 1, rw, acc, a, b0, b1, lane, wave);
    } else {
      ktile2<NS, SUM, 0>(smem, oa, ob, f, lds_stage, w, kt, rw, acc, a, b0, b1, lane, wave);
      if (kt + 1 < w.nk) ktile2<NS, SUM, 1>(smem, oa, ob, f, lds_stage, w, kt + 1, rw, acc, a, b0, b1, lane, wave);
    }
  }
  if (wr == 0) G256W_BARRIER();

  const int row0 = wr * 64 + (lane & 15), col0 = wc * 32 + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epi((i >> 2) * 128 + (i & 3) * 16 + row0, (j >> 1) * 128 + (j & 1) * 16 + col0, acc[i][j]);
  if constexpr (SUM) {      // the two row halves of a column meet
    __syncthreads();
    const float* sums = reinterpret_cast<const float*>(smem + 2 * BUF);
    if (threadIdx.x < 256) episum((int)threadIdx.x, sums[threadIdx.x] + sums[threadIdx.x + 256]);
  }
}

}  // namespace g256w
