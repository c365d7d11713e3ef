// The layer chain (gemm_bf16_chain.hpp): the queue behind nacf_chain_begin / nacf_chain_flush and the one persistent launch.
#undef NACF_GEMM_TRACE
#undef NACF_BF16_TRACE
#include <mutex>
#include "gemm_bf16_launch.hpp"
#include "gemm_bf16_chain.hpp"

namespace {
// The queue is per host thread (ADVICE round 3 on the wide group's process-global queue: a GEMM issued by another thread
// -- autograd's, a side stream's -- must not land in somebody else's chain); the device-side barrier counters are per
// device, which is why chains of different streams must not overlap (gemm_bf16_chain.hpp).
thread_local bool t_on = false;
thread_local chain::Table t_tab;
thread_local int t_nlin = 0, t_natt = 0;
thread_local char t_last[160] = "";

int n_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
void reset() { t_tab.n = 0; t_nlin = 0; t_natt = 0; }

using G = panel::Geo<2, 4, 3>;
int launch(hipStream_t s) {
  if (t_tab.n == 0) return 0;
  // NACF_CHAIN_FENCE (tuning, read per launch): the cache maintenance of the grid barrier, see gemm_bf16_chain.hpp
  const char* fe = getenv("NACF_CHAIN_FENCE");
  const int fence = fe ? atoi(fe) : 1;
  void (*kern)(chain::Table) = fence == 0 ? chain::chain_kernel<2, 4, 0> : fence == 2 ? chain::chain_kernel<2, 4, 2> : chain::chain_kernel<2, 4, 1>;
  static bool raised[3] = {false, false, false};
  if (!raised[fence == 0 ? 0 : fence == 2 ? 2 : 1]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    raised[fence == 0 ? 0 : fence == 2 ? 2 : 1] = true;
  }
  // one workgroup per CU, a multiple of 8 (bx & 7 = the XCD); never more than the CUs (the grid barrier needs them co-resident)
  const int grid = n_cus() / 8 * 8;
  { const char* te = getenv("NACF_CHAIN_TRACE"); t_tab.trace = te ? atoi(te) : 0; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, s, t_tab);
  const int n = t_tab.n;
  snprintf(t_last, sizeof(t_last), "chain_kernel<2, 4>[%d stages: %d linear, %d attention]", n, t_nlin, t_natt);
  bf16_note_wide(t_last);
  reset();
  return n;
}
}  // namespace

bool chain_active() { return t_on; }

bool chain_queue_linear(const GemmShape& g0, const EpiLinear& epi, hipStream_t s) {
  if (!t_on || !panel_eligible(g0)) return false;
  if (t_nlin == chain::MAX_LINEAR) launch(s);         // full: what is queued runs first (stream order keeps the dependence)
  GemmShape g = g0;
  g.tiles_m = cdiv(g.M, G::BM);
  g.tiles_n = g.N / G::BN;
  t_tab.kind[t_tab.n] = chain::ST_LINEAR;
  t_tab.idx[t_tab.n] = (unsigned char)t_nlin;
  t_tab.g[t_nlin] = g;
  t_tab.epi[t_nlin] = epi;
  ++t_nlin; ++t_tab.n;
  bf16_note_wide("chain_queued");
  return true;
}

bool chain_queue_attention(const float* Q, int64_t ldq, const float* K, int64_t ldk, const float* V, int64_t ldv, float* O, int64_t ldo,
                           const int64_t* key_tokens, int causal, float* probs, int R, int H, int Lq, int Lk, int dk, int kv_div,
                           int kv_mod, bool aligned, hipStream_t s) {
  // the matrix-core forward with 64-wide heads only (attn::fwd_item<2 | 8, 4>, attn::fwd_lds_item<4>): the same choice between
  // the streaming and the LDS-staged form as nacf_attention_fwd makes
  if (!t_on || !aligned || dk != 64 || Lk > 128 || (causal ? Lq : (Lq < 32 ? Lq : 32)) > 32) return false;
  if (t_natt == chain::MAX_ATTN) launch(s);
  chain::AttnArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.key_tokens = key_tokens; a.probs = probs; a.causal = causal; a.R = R; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.kv_div = kv_div; a.kv_mod = kv_mod; a.nqb = cdiv(Lq, 32); a.nkt = Lk <= 32 ? 2 : 8;
  const int groups = cdiv(R, kv_div);
  a.n_kv = kv_mod < groups ? kv_mod : groups;
  a.rounds = kv_div * cdiv(groups, kv_mod);
  const char* e = getenv("NACF_ATTN_LDS");
  const bool lds = Lk > 32 && !key_tokens && !causal && !probs && a.rounds * a.nqb >= 4 && !(e && atoi(e) == 0);
  t_tab.kind[t_tab.n] = lds ? chain::ST_ATTN_LDS : chain::ST_ATTN;
  t_tab.idx[t_tab.n] = (unsigned char)t_natt;
  t_tab.at[t_natt] = a;
  ++t_natt; ++t_tab.n;
  return true;
}

int chain_flush(hipStream_t s) { return launch(s); }

extern "C" {
int nacf_chain_begin(void) {
  t_on = true;
  reset();
  return NACF_OK;
}
int nacf_chain_flush(nacf_stream_t stream) {
  t_on = false;
  const int n = launch(as_hip(stream));
  NACF_LAUNCH_CHECK("nacf_chain_flush");
  return n;      // >= 0: stages the flush launched
}
int nacf_chain_status(nacf_stream_t stream) {
  // synchronises `stream`; 1 = a workgroup of some chain launch gave up waiting at a grid barrier since the last call
  unsigned flags[4] = {0, 0, 0, 0};
  if (hipStreamSynchronize(as_hip(stream)) != hipSuccess ||
      hipMemcpyFromSymbol(flags, HIP_SYMBOL(chain::g_chain_sync), sizeof(flags)) != hipSuccess) {
    nacf_set_error("nacf_chain_status: cannot read the device flags");
    return NACF_ELAUNCH;
  }
  if (flags[2] != 0 || flags[0] != 0 || flags[1] != 0) {      // (nonzero counters after a finished launch: it did not leave cleanly)
    const unsigned zero[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(chain::g_chain_sync), zero, sizeof(zero));
    return 1;
  }
  return 0;
}
int nacf_chain_stamps(uint64_t* out, int n, nacf_stream_t stream) {
  // tuning (NACF_CHAIN_TRACE=1): the stage stamps of the last chain launch, see gemm_bf16_chain.hpp:g_chain_stamp
  NACF_CHECK(out && n > 0 && n <= 3 * chain::MAX_STAGES + 1, NACF_EINVAL, "nacf_chain_stamps: bad argument");
  if (hipStreamSynchronize(as_hip(stream)) != hipSuccess ||
      hipMemcpyFromSymbol(out, HIP_SYMBOL(chain::g_chain_stamp), sizeof(uint64_t) * n) != hipSuccess) {
    nacf_set_error("nacf_chain_stamps: cannot read the device stamps");
    return NACF_ELAUNCH;
  }
  return NACF_OK;
}
}
