// bf16-RESIDENT matrix-core GEMM body: 256 x 256 x 64 workgroup tile, eight phases per pair of k-tiles, both operands
// global -> LDS by DMA (buffer_load_dwordx4 ... lds), counted vmcnt, raw s_barrier, the two wave rows one barrier apart.
// Schedule after /opt/skills/guides/cdna_hip_programming.md, "The 256^2 8-phase template" (geometry table, k-loop block, the
// vmcnt / read / restage placement rules); measured 1.35-1.41 PF on 8192^3, 1.14-1.31 PF on 4096^3 with random operands
// (profiles/r05_gemm256_8phase_probe.txt; the 128 x 128 two-barrier body of round 3: 0.82-0.86 PF).
//
// Two operand layouts behind one k-loop:
//   F form (forward / dX):  C[m][n] = sum_k A[m][k] * B[n][k]      A, B bf16, k contiguous           fragments: ds_read_b128
//   W form (dW):            C[i][j] = sum_r A[r][i] * B[r][j]      A, B bf16, the REDUCE index is the row   fragments: 2 x ds_read_b64_tr_b16
//
//   512 threads = 8 waves as 2 (wr) x 4 (wc).  LDS 128 KiB = 2 k-tiles x {A0, A1, B0, B1} half-tiles of 16 KiB:
//     F form: half h of A = tile rows h*128 .. +127, 64 k each (128-byte LDS rows, 16-byte chunks XOR-swizzled by row & 7)
//     W form: half h of A = tile columns h*128 .. +127, 64 reduce rows each (256-byte LDS rows, 32-byte blocks XOR-swizzled
//             by (r & 3) | ((r >> 3) & 1) << 2: the eight rows a 32-lane group of a transposing read touches land on eight
//             different blocks of the bank row)
//   The image of an LDS-DMA is lane-linear, so the swizzle is applied to the SOURCE address and again on the read.
//   Wave (wr, wc) owns rows {h*128 + wr*64 + 0..63} x columns {g*128 + wc*32 + 0..31}, h, g = 0, 1: every wave reads both halves
//   of both operands, one quadrant (h, g) = 64 x 32 outputs = 16 v_mfma_f32_16x16x32_bf16 per phase:
//
//     phase 1: read B0, A0      stage A1[t+1]   lgkmcnt(retire B0) | barrier | A0 x B0 | barrier
//     phase 2: read B1          stage B0[t+2]                      | barrier | A0 x B1 | barrier
//     phase 3: read A1          stage A0[t+2]                      | barrier | A1 x B1 | barrier
//     phase 4:                  stage B1[t+2]   vmcnt(6)           | barrier | A1 x B0 | barrier
//
//   vmcnt(6) leaves the three newest half-tiles in flight and retires tile t+1 completely: the k-loop never waits for zero.
//   A buffer is read one phase after the wait + barrier that retires it and restaged two phases after its last read (B0 one
//   phase after: its reads are retired ahead of phase 1's first barrier).  Rows of an operand beyond its extent are read as
//   zeros (descriptor bounds); the k extent must be padded with zeros in at least one operand (F form: to a multiple of 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace g256 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) bf16x4* lds_b4;

constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
constexpr int HALF = 128 * 128;            // bytes of a half-tile
constexpr int BUF = 4 * HALF;              // one k-tile: A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF;

// a pointer the compiler cannot prove wave-uniform (it is): through readfirstlane, or every buffer operation built on it is
// wrapped in a waterfall loop (cdna_hip_programming.md, T20)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* q) {
  const uint64_t v = reinterpret_cast<uint64_t>(q);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// one DMA-staged operand as a workgroup sees it: byte offset of (half h, request j, k-tile kt) = base + h*half_step + j*j_step + kt*kt_step
struct Stage {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff;                           // per lane
  unsigned base, half_step, j_step, kt_step;
};

// F form: the operand is [rows][ld] bf16, k contiguous; the tile starts at row0; rows >= rows_total read as zeros.
__device__ __forceinline__ Stage stage_f(const uint16_t* p, int64_t ld, int row0, int rows_total, int kt0, int lane, int wave) {
  Stage s;
  const size_t bytes = (size_t)(rows_total > row0 ? rows_total - row0 : 0) * (size_t)ld * 2;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(p + (size_t)row0 * ld), 0,
                                             (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xfffffff0u ? bytes : 0xfffffff0u)), 0x00020000);
  const unsigned rowstep = (unsigned)ld * 2u;
  s.voff = (unsigned)(lane >> 3) * rowstep + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
  s.base = (unsigned)(wave * 16) * rowstep + (unsigned)kt0 * 128u;
  s.half_step = 128u * rowstep;
  s.j_step = 8u * rowstep;
  s.kt_step = 128u;
  return s;
}
// W form: the operand is [rows][ld] bf16, the reduce index is the row; the tile starts at column col0; rows >= rows_total read as zeros.
__device__ __forceinline__ Stage stage_w(const uint16_t* p, int64_t ld, int col0, int rows_total, int kt0, int lane, int wave) {
  Stage s;
  const size_t all = (size_t)rows_total * (size_t)ld * 2, off = (size_t)col0 * 2;
  const size_t bytes = all > off ? all - off : 0;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(p + col0), 0,
                                             (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xfffffff0u ? bytes : 0xfffffff0u)), 0x00020000);
  const unsigned rowstep = (unsigned)ld * 2u;
  const int r4 = lane >> 4, c16 = lane & 15;
  s.voff = (unsigned)r4 * rowstep + (unsigned)((((c16 >> 1) ^ (r4 | ((wave & 1) << 2))) << 5) + ((c16 & 1) << 4));
  s.base = (unsigned)(kt0 * 64 + wave * 8) * rowstep;
  s.half_step = 256u;
  s.j_step = 4u * rowstep;
  s.kt_step = 64u * rowstep;
  return s;
}

// per-lane LDS addresses of the fragment reads
struct Frag {
  unsigned a[4], b[4];      // F form: a[ks], b[ks] (ks = 0, 1).  W form: a[mf] (mf = 0..3), b[nf] (nf = 0, 1)
};
template <bool WFORM>
__device__ __forceinline__ Frag frag_addr(int lane, int wr, int wc) {
  Frag f;
  if constexpr (!WFORM) {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned chunk = (unsigned)(((ks * 4 + q) ^ (lane & 7)) << 4);
      f.a[ks] = (unsigned)(wr * 64 + r) * 128u + chunk;
      f.b[ks] = (unsigned)(wc * 32 + r) * 128u + chunk;
    }
    f.a[2] = f.a[3] = f.b[2] = f.b[3] = 0;
  } else {
    const int g = lane >> 4, i16 = lane & 15;
    const unsigned row = (unsigned)(g * 8 + (i16 >> 2)) * 256u + (unsigned)((i16 & 3) * 8);
    const int s = (i16 >> 2) | ((g & 1) << 2);
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) f.a[mf] = row + (unsigned)(((wr * 4 + mf) ^ s) << 5);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) f.b[nf] = row + (unsigned)(((wc * 2 + nf) ^ s) << 5);
    f.b[2] = f.b[3] = 0;
  }
  return f;
}

template <int BUFI, int OPND, int H>
__device__ __forceinline__ void stage(unsigned char* smem, const Stage& s, unsigned lds_stage, int kt) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (lds_ptr)(smem + BUFI * BUF + OPND * 2 * HALF + H * HALF + lds_stage + j * 1024), 16, s.voff,
                                             s.base + H * s.half_step + j * s.j_step + kt * s.kt_step, 0, 0);
}
template <bool WFORM, int BUFI, int H>
__device__ __forceinline__ void read_a(const unsigned char* smem, const Frag& f, bf16x8 (&a)[4][2]) {
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (!WFORM) {
        a[mf][ks] = *reinterpret_cast<const bf16x8*>(smem + BUFI * BUF + H * HALF + mf * 2048 + f.a[ks]);
      } else {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(smem + BUFI * BUF + H * HALF + ks * 8192 + f.a[mf]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(smem + BUFI * BUF + H * HALF + ks * 8192 + 1024 + f.a[mf]));
        a[mf][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
}
template <bool WFORM, int BUFI, int G>
__device__ __forceinline__ void read_b(const unsigned char* smem, const Frag& f, bf16x8 (&b)[2][2]) {
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (!WFORM) {
        b[nf][ks] = *reinterpret_cast<const bf16x8*>(smem + BUFI * BUF + 2 * HALF + G * HALF + nf * 2048 + f.b[ks]);
      } else {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(smem + BUFI * BUF + 2 * HALF + G * HALF + ks * 8192 + f.b[nf]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(smem + BUFI * BUF + 2 * HALF + G * HALF + ks * 8192 + 1024 + f.b[nf]));
        b[nf][ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
}
template <int MH, int NH>
__device__ __forceinline__ void quadrant(f32x4 (&acc)[8][4], const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        acc[MH * 4 + mf][NH * 2 + nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][ks], a[mf][ks], acc[MH * 4 + mf][NH * 2 + nf], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
}

#define G256_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define G256_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// one k-tile = four phases; kt counts from the workgroup's first k-tile, nk = its number of k-tiles
template <bool WFORM, int BUFI>
__device__ __forceinline__ void ktile(unsigned char* smem, const Stage& sa, const Stage& sb, const Frag& f, unsigned lds_stage, int kt, int nk,
                                      f32x4 (&acc)[8][4], bf16x8 (&a)[4][2], bf16x8 (&b0)[2][2], bf16x8 (&b1)[2][2]) {
  const bool next1 = kt + 1 < nk, next2 = kt + 2 < nk;      // wave-uniform
  // ---- phase 1
  read_b<WFORM, BUFI, 0>(smem, f, b0);
  __builtin_amdgcn_sched_barrier(0);
  read_a<WFORM, BUFI, 0>(smem, f, a);
  if (next1) stage<BUFI ^ 1, 0, 1>(smem, sa, lds_stage, kt + 1);
  // B0's reads retired before the first barrier: B0 is restaged in the next phase (the counter holds 15 at most: W form)
  if constexpr (WFORM) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  G256_BARRIER();
  G256_LGKM0();
  quadrant<0, 0>(acc, a, b0);
  G256_BARRIER();
  // ---- phase 2
  read_b<WFORM, BUFI, 1>(smem, f, b1);
  if (next2) stage<BUFI, 1, 0>(smem, sb, lds_stage, kt + 2);
  G256_BARRIER();
  G256_LGKM0();
  quadrant<0, 1>(acc, a, b1);
  G256_BARRIER();
  // ---- phase 3
  read_a<WFORM, BUFI, 1>(smem, f, a);
  if (next2) stage<BUFI, 0, 0>(smem, sa, lds_stage, kt + 2);
  G256_BARRIER();
  G256_LGKM0();
  quadrant<1, 1>(acc, a, b1);
  G256_BARRIER();
  // ---- phase 4
  if (next2) stage<BUFI, 1, 1>(smem, sb, lds_stage, kt + 2);
  if (next2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G256_BARRIER();
  quadrant<1, 0>(acc, a, b0);
  G256_BARRIER();
}

// The whole workgroup: prologue, k-loop over nk (>= 1) k-tiles, epilogue through `epi(row, col, v)`: v = the four outputs
// (row, col .. col + 3) of the 256 x 256 tile, row / col tile-relative.  sa / sb already point at the workgroup's first k-tile.
template <bool WFORM, class Epi>
__device__ __forceinline__ void body(unsigned char* smem, const Stage& sa, const Stage& sb, int nk, const Epi& epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const Frag f = frag_addr<WFORM>(lane, wr, wc);
  const unsigned lds_stage = (unsigned)wave * 2048u;      // 16 F-form rows of 128 bytes = 8 W-form rows of 256 bytes

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4][2], b0[2][2], b1[2][2];

  // ---- prologue: all of tile 0, then tile 1's B0, A0, B1 (its A1 is staged by tile 0's phase 1)
  stage<0, 0, 0>(smem, sa, lds_stage, 0);
  stage<0, 1, 0>(smem, sb, lds_stage, 0);
  stage<0, 1, 1>(smem, sb, lds_stage, 0);
  stage<0, 0, 1>(smem, sa, lds_stage, 0);
  if (nk > 1) {
    stage<1, 1, 0>(smem, sb, lds_stage, 1);
    stage<1, 0, 0>(smem, sa, lds_stage, 1);
    stage<1, 1, 1>(smem, sb, lds_stage, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G256_BARRIER();
  if (wr == 1) G256_BARRIER();      // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < nk; kt += 2) {
    ktile<WFORM, 0>(smem, sa, sb, f, lds_stage, kt, nk, acc, a, b0, b1);
    if (kt + 1 < nk) ktile<WFORM, 1>(smem, sa, sb, f, lds_stage, kt + 1, nk, acc, a, b0, b1);
  }
  if (wr == 0) G256_BARRIER();

  // ---- epilogue: acc[h*4 + mf][g*2 + nf] at lane l: row h*128 + wr*64 + mf*16 + (l & 15), columns g*128 + wc*32 + nf*16 + (l >> 4)*4 + 0..3
  const int row0 = wr * 64 + (lane & 15), col0 = wc * 32 + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epi((i >> 2) * 128 + (i & 3) * 16 + row0, (j >> 1) * 128 + (j & 1) * 16 + col0, acc[i][j]);
}

// bijective XCD remap (block b runs on XCD b % 8: each XCD gets a contiguous run of logical tiles), then strips of group_m tile rows
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int per_group = group_m * tiles_n, grp = logical / per_group, in_grp = logical - grp * per_group;
  const int rows_here = min(group_m, tiles_m - grp * group_m);
  tm = grp * group_m + in_grp % rows_here;
  tn = in_grp / rows_here;
}

}  // namespace g256
