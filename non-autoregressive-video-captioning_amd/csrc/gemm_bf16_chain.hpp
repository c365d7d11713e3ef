// The layer chain: ONE persistent launch that runs a whole decoder layer's forward pass (models/bert.py:262-303:
// q|k|v -> self-attention -> output projection + residual -> cross-attention query -> cross-attention -> output projection
// + residual -> FFN1 -> FFN2 + residual), i.e. a LIST of dependent stages, instead of one launch per nn.Linear / attention
// core.  Launched one at a time these are 8-10 kernels of 15-35 us, each of which spends a third of its life getting
// started (arguments -> live-row count -> row list -> first operands) and draining (partly filled last round of
// workgroups); here a grid of one workgroup per CU stays resident and walks the stages:
//
//   stage kinds   LINEAR    the panel GEMM body (gemm_bf16_panel.hpp: 64 x 128 tiles, k-split waves, weights streamed from the
//                           fragment-major images into the accumulation file) with the fused nn.Linear epilogue, every
//                           workgroup its contiguous share of the stage's tiles;
//                 ATTN      attn::fwd_item (attn_mfma.hpp): one wave per (sequence, head, 32 queries), operands straight
//                           from L2 into the matrix instruction's fragments -- items dealt to the grid's waves round robin;
//                 ATTN_LDS  attn::fwd_lds_item: one workgroup per (memory row set, head), K / V staged in LDS once for every
//                           sequence that attends to it (the six length candidates of a video in NA decoding).
//   between two stages a GRID BARRIER: each wave releases its stores (agent scope: the L2s of the eight XCDs are not coherent
//   with each other for plain stores), the workgroup arrives on a device-wide counter, spins until all have, and acquires.
//   The counter is monotonic inside a launch (stage s waits for (s + 1) * grid arrivals) and is reset by the last workgroup to
//   leave the launch, so the next launch (stream order) finds zeros; a workgroup that spins longer than ~0.25 s gives up,
//   raises g_chain_sync[2] and lets the launch end (the results are then wrong, the GPU is not hung: the host checks the
//   flag in nacf_chain_status).  All workgroups must be co-resident: the grid is at most the number of CUs and the kernel's
//   128 KB of LDS lets no second workgroup onto a CU, so this holds whenever the launch has the device to itself -- one
//   stream, which is how the runtime (and a captured hipGraph) issues it.
//
// Arithmetic and summation order of every stage are those of the stand-alone kernels (the same device functions), so a chained
// layer is bit-identical to the same layer launched stage by stage with the panel kernel forced on (tests/test_chain_gpu.py).
#pragma once
#include "gemm_bf16_panel.hpp"
#include "attn_mfma.hpp"

namespace chain {
constexpr int MAX_LINEAR = 8, MAX_ATTN = 4, MAX_STAGES = MAX_LINEAR + MAX_ATTN;
enum : int { ST_LINEAR = 0, ST_ATTN = 1, ST_ATTN_LDS = 2 };

struct AttnArgs {
  const float* Q; const float* K; const float* V; float* O;
  int64_t ldq, ldk, ldv, ldo;
  const int64_t* key_tokens;
  float* probs;
  int causal, R, H, Lq, Lk, kv_div, kv_mod, nqb;
  int nkt;           // ATTN: key tiles of 16 the instantiation covers (2 or 8)
  int n_kv, rounds;  // ATTN_LDS: memory row sets, sequences per row set (upper bound)
  int pad_;
};

struct Table {
  int n;
  int trace;         // != 0: wall-clock stamps per stage into g_chain_stamp (nacf_chain_stamps)
  unsigned char kind[MAX_STAGES];
  unsigned char idx[MAX_STAGES];
  GemmShape g[MAX_LINEAR];
  EpiLinear epi[MAX_LINEAR];
  AttnArgs at[MAX_ATTN];
};
static_assert(sizeof(Table) <= 3800, "the table travels by value in the kernel arguments (4 KB)");

// [0] barrier arrivals, [1] exit arrivals, [2] != 0: some workgroup gave up waiting (sticky; nacf_chain_status reads and clears it)
__device__ unsigned g_chain_sync[4];
// tuning (Table::trace): per stage s, 100 MHz wall clock: [3 s] workgroup 0 starts the stage, [3 s + 1] workgroup 0 has done its
// share, [3 s + 2] the last workgroup has (atomic max); [3 n] workgroup 0 is past the last stage
__device__ unsigned long long g_chain_stamp[3 * MAX_STAGES + 1];

// FENCE: 0 = every wave releases / acquires at agent scope (buffer_wbl2 sc1 / buffer_inv sc1 from 1024 waves per barrier);
//        1 = every wave waits for its own stores, ONE wave per workgroup writes the L2 back before it arrives and invalidates
//            after the barrier (the caches are per CU / per XCD, not per wave);
//        2 = no cache maintenance at all (tuning only: what the barrier itself costs -- results are NOT defined)
template <int FENCE>
__device__ __forceinline__ void grid_barrier(const unsigned target) {
  if constexpr (FENCE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if constexpr (FENCE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&g_chain_sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__hip_atomic_load(&g_chain_sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 25000000ull) {              // 0.25 s: not co-resident, or a workgroup died
        __hip_atomic_store(&g_chain_sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if constexpr (FENCE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if constexpr (FENCE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MT, int NT, int FENCE>
__global__ __launch_bounds__(256, 1) void chain_kernel(Table tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int bx = (int)blockIdx.x, Gx = (int)gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#pragma unroll 1
  for (int s = 0; s < tab.n; ++s) {
    const int kind = tab.kind[s], j = tab.idx[s];
    if (tab.trace && threadIdx.x == 0 && bx == 0) g_chain_stamp[3 * s] = __builtin_amdgcn_s_memrealtime();
    if (kind == ST_LINEAR) {
      panel::panel_body<MT, NT, 3, EpiLinear>(tab.g[j], tab.epi[j], bx, Gx, 0, smem_raw);
    } else if (kind == ST_ATTN) {
      const AttnArgs& a = tab.at[j];
      const int n_items = a.R * a.H * a.nqb;
      // the blocks of one (sequence, head) are consecutive items = the waves of one workgroup (they share K / V rows)
#pragma unroll 1
      for (int item = 4 * bx + wave; item < n_items; item += 4 * Gx) {
        if (a.nkt == 2)
          attn::fwd_item<2, 4>(item, a.Q, a.ldq, a.K, a.ldk, a.V, a.ldv, a.O, a.ldo, a.key_tokens, a.causal, a.probs, a.R, a.H, a.Lq,
                               a.Lk, a.kv_div, a.kv_mod, a.nqb);
        else
          attn::fwd_item<8, 4>(item, a.Q, a.ldq, a.K, a.ldk, a.V, a.ldv, a.O, a.ldo, a.key_tokens, a.causal, a.probs, a.R, a.H, a.Lq,
                               a.Lk, a.kv_div, a.kv_mod, a.nqb);
      }
    } else {
      const AttnArgs& a = tab.at[j];
      const int n_items = a.n_kv * a.H;
#pragma unroll 1
      for (int item = bx; item < n_items; item += Gx) {
        attn::fwd_lds_item<4>(item, reinterpret_cast<float*>(smem_raw), a.Q, a.ldq, a.K, a.ldk, a.V, a.ldv, a.O, a.ldo, a.R, a.H, a.Lq,
                              a.Lk, a.kv_div, a.kv_mod, a.rounds);
        __syncthreads();        // the next item's copy overwrites K / V
      }
    }
    if (tab.trace && threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      if (bx == 0) g_chain_stamp[3 * s + 1] = t;
      atomicMax(&g_chain_stamp[3 * s + 2], t);
    }
    if (s + 1 < tab.n) grid_barrier<FENCE>((unsigned)(s + 1) * (unsigned)Gx);
  }
  if (tab.trace && threadIdx.x == 0 && bx == 0) g_chain_stamp[3 * tab.n] = __builtin_amdgcn_s_memrealtime();
  // leave: the last workgroup out resets the counters for the next launch
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned k = __hip_atomic_fetch_add(&g_chain_sync[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (k + 1 == (unsigned)Gx) {
      __hip_atomic_store(&g_chain_sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g_chain_sync[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
}  // namespace chain
