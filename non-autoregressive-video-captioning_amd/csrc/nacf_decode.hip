// Device-side bookkeeping of the NA decoding loop (decoding/na_generate.py,
// decoding/algorithms.py): length beam, canvas, select_worst re-masking and the
// final best-candidate pick.  Integer / index work: bit-exact vs the reference.
// These replace per-row Python loops and host syncs, not FLOPs.
#include "common.hpp"

namespace {

__global__ void length_beam_kernel(const float* __restrict__ pred_length, int B, int max_len, int lbs, int bias,
                                   int32_t* __restrict__ beam, int32_t* __restrict__ beam_max) {
  __shared__ int red[256];
  int local_max = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* p = pred_length + (int64_t)b * max_len;
    unsigned long long taken = 0ull;  // max_len <= 64
    for (int j = 0; j < lbs; ++j) {
      float best = -3.0e38f;
      int bi = -1;
#pragma unroll 4      // the load is unconditional: behind `continue` every one of the lbs x max_len loads waited for the one before it
      for (int i = 0; i < max_len; ++i) {
        const float v = p[i];
        if (!((taken >> i) & 1ull) && (bi < 0 || v > best)) { best = v; bi = i; }
      }
      taken |= 1ull << bi;
      int len = bi + bias;
      if (len < 4) len = 4;                       // decoding/na_generate.py:130-132
      if (len > max_len - 1) len = max_len - 1;
      beam[(int64_t)b * lbs + j] = len;
      local_max = max(local_max, len);
    }
  }
  red[threadIdx.x] = local_max;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) beam_max[0] = red[0];
}

// gold-length beam (opt['load_generated_captions'], decoding/na_generate.py:25-26,118-122): lbs consecutive lengths around the
// number of non-<pad> tokens of the given caption, then the same clamp as the predicted beam
__global__ void length_beam_gold_kernel(const int64_t* __restrict__ tgt_tokens, int B, int T, int max_len, int lbs,
                                        int32_t* __restrict__ beam, int32_t* __restrict__ beam_max) {
  __shared__ int red[256];
  int local_max = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int64_t* t = tgt_tokens + (int64_t)b * T;
    int gold = 0;
    for (int i = 0; i < T; ++i) gold += (t[i] != NACF_PAD) ? 1 : 0;
    const int start = gold - (lbs - 1) / 2;
    for (int j = 0; j < lbs; ++j) {
      int len = start + j;
      if (len < 4) len = 4;                       // decoding/na_generate.py:130-132
      if (len > max_len - 1) len = max_len - 1;
      beam[(int64_t)b * lbs + j] = len;
      local_max = max(local_max, len);
    }
  }
  red[threadIdx.x] = local_max;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) beam_max[0] = red[0];
}

__global__ void canvas_init_kernel(const int32_t* __restrict__ beam, int rows, int Lp, int64_t* __restrict__ tokens) {
  const int64_t total = (int64_t)rows * Lp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Lp), l = (int)(e % Lp);
    tokens[e] = l < beam[r] ? NACF_MASK : NACF_PAD;
  }
}

// the canvas of opt['load_generated_captions'] (decoding/na_generate.py:42-50): candidate (b, j) starts from caption b with
// <pad> -> <mask>, cut to its candidate length
__global__ void canvas_init_gold_kernel(const int32_t* __restrict__ beam, const int64_t* __restrict__ tgt_tokens, int T, int rows,
                                        int lbs, int Lp, int64_t* __restrict__ tokens) {
  const int64_t total = (int64_t)rows * Lp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Lp), l = (int)(e % Lp);
    int64_t tok = NACF_PAD;
    if (l < beam[r]) {
      tok = l < T ? tgt_tokens[(int64_t)(r / lbs) * T + l] : (int64_t)NACF_MASK;
      if (tok == NACF_PAD) tok = NACF_MASK;
    }
    tokens[e] = tok;
  }
}

// one wave per row, one lane per position (Lp <= 64)
__global__ void select_mask_kernel(const float* __restrict__ probs, const float* __restrict__ teacher,
                                   const int64_t* __restrict__ pad_tokens, const int32_t* __restrict__ lut, int mode,
                                   int rows, int Lp, int64_t* __restrict__ tokens, uint8_t* __restrict__ mask_out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bool in = lane < Lp;
  const int64_t o = (int64_t)row * Lp + lane;
  const bool is_pad = in ? (pad_tokens[o] == NACF_PAD) : true;
  bool sel = false;
  if (mode == 1) {
    sel = in && tokens[o] == NACF_MASK;
  } else if (mode == 2) {
    sel = in && tokens[o] != NACF_MASK && !is_pad;
  } else {
    const int seq_len = __popcll(__ballot(in && !is_pad));
    int n = lut[seq_len];
    if (n < 1) n = 1;                              // decoding/algorithms.py:213
    float s = 3.0e38f;
    if (in) s = probs[o] * (teacher ? teacher[o] : 1.f);
    int rank = 0;
    for (int j = 0; j < Lp; ++j) {
      const float sj = __shfl(s, j, 64);
      rank += (sj < s || (sj == s && j < lane)) ? 1 : 0;
    }
    sel = in && rank < n;
  }
  if (in) {
    if (sel) tokens[o] = NACF_MASK;
    mask_out[o] = sel ? 1 : 0;
  }
}

// one wave per video: score every length candidate and copy the best
__global__ void best_candidate_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ probs,
                                      const float* __restrict__ teacher, const int32_t* __restrict__ beam, float alpha,
                                      int B, int lbs, int Lp, int64_t* __restrict__ out_tokens,
                                      int32_t* __restrict__ best_idx, float* __restrict__ cand_lprobs) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  float best = 0.f;
  int bj = -1;
  for (int j = 0; j < lbs; ++j) {
    const int64_t row = (int64_t)b * lbs + j;
    float lp = 0.f;
    if (lane < Lp) {
      const int64_t o = row * Lp + lane;
      lp = logf(probs[o] * (teacher ? teacher[o] : 1.f));
      if (cand_lprobs) cand_lprobs[o] = lp;
    }
    const float tot = wave_sum(lp);
    const float sc = tot / powf((float)beam[row], alpha);  // decoding/na_generate.py:72
    if (bj < 0 || sc > best) { best = sc; bj = j; }
  }
  if (lane < Lp) out_tokens[(int64_t)b * Lp + lane] = tokens[((int64_t)b * lbs + bj) * Lp + lane];
  if (lane == 0 && best_idx) best_idx[b] = bj;
}

__global__ void token_replace_kernel(int64_t* __restrict__ tokens, int64_t n, int64_t from, int64_t to) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (tokens[e] == from) tokens[e] = to;
}
__global__ void teacher_probs_kernel(const float* __restrict__ label_logp, const int64_t* __restrict__ pad_tokens,
                                     float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = pad_tokens[e] == NACF_PAD ? 1.0f : expf(label_logp[e]);
}
__global__ void init_probs_kernel(const int64_t* __restrict__ pad_tokens, float* __restrict__ probs, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    probs[e] = pad_tokens[e] == NACF_PAD ? 1.0f : 0.0f;
}
__global__ void apply_mask_kernel(int64_t* __restrict__ tokens, const uint8_t* __restrict__ mask, int64_t value,
                                  int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (mask[e]) tokens[e] = value;
}

// single workgroup: rank of each <mask> slot inside its row, max per-row count, total count
__global__ void mask_rank_kernel(const int64_t* __restrict__ tokens, int rows, int Lp, int32_t* __restrict__ rank,
                                 int32_t* __restrict__ counts) {
  __shared__ int red_max[4];
  __shared__ int red_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int lmax = 0, ltot = 0;
  for (int row = wave; row < rows; row += 4) {
    const bool in = lane < Lp;
    const bool m = in && tokens[(int64_t)row * Lp + lane] == NACF_MASK;
    const unsigned long long bal = __ballot(m);
    if (in) rank[(int64_t)row * Lp + lane] = m ? __popcll(bal & ((1ull << lane) - 1ull)) : -1;
    const int c = __popcll(bal);
    lmax = max(lmax, c);
    ltot += c;
  }
  if (lane == 0) { red_max[wave] = lmax; red_tot[wave] = ltot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[0] = max(max(red_max[0], red_max[1]), max(red_max[2], red_max[3]));
    counts[1] = red_tot[0] + red_tot[1] + red_tot[2] + red_tot[3];
  }
}
__global__ void select_rank_kernel(const int32_t* __restrict__ rank, int cur, int q, int64_t n,
                                   int64_t* __restrict__ tokens, uint8_t* __restrict__ mask_out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank[e];
    const bool sel = r >= cur && r < cur + q;
    if (sel) tokens[e] = NACF_MASK;
    mask_out[e] = sel ? 1 : 0;
  }
}
// one wave per row: among <mask> slots take the min(q, remaining) most confident new predictions
__global__ void easy_first_update_kernel(int64_t* __restrict__ tokens, float* __restrict__ probs,
                                         const int64_t* __restrict__ new_tokens, const float* __restrict__ new_probs,
                                         int q, int rows, int Lp) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bool in = lane < Lp;
  const int64_t o = (int64_t)row * Lp + lane;
  const bool m = in && tokens[o] == NACF_MASK;
  const int remain = __popcll(__ballot(m));
  if (remain == 0) return;
  const int k = min(q, remain);
  const float s = m ? new_probs[o] : 0.0f;          // token_probs[~mask_ind] = 0 (algorithms.py:373)
  int rank = 0;
  for (int j = 0; j < Lp; ++j) {
    const float sj = __shfl(s, j, 64);
    rank += (sj > s || (sj == s && j < lane)) ? 1 : 0;
  }
  if (in && rank < k) { tokens[o] = new_tokens[o]; probs[o] = s; }
}

// ---------------------------------------------------------------- AR beam search step
constexpr int BEAM_MAX = 8;

// One workgroup per instance: Beam.advance (models/Beam.py:68-117) for step t.  NB = beam size (compile time): every thread
// keeps the NB best of its candidates in a sorted register list, BRANCH-FREE -- a candidate bubbles through the list with
// compare + select pairs, wanted or not.  (With the beam size a run-time bound and the insertion behind a "better than my
// worst" test, hipcc built ~330 branches and 1350 register moves out of it; in a wave some lane always wants the insertion,
// so every candidate paid for all of it: 173 us per step at B = 256 with 256 threads, 74 with 1024; the 54 MB read are worth ~15.)
constexpr int BEAM_THREADS = 1024;
template <int NB>
__global__ __launch_bounds__(BEAM_THREADS) void beam_step_kernel(const float* __restrict__ logp, int64_t ldl, int V,
                                                         int t, int max_len, int want, int64_t* __restrict__ seqs,
                                                         float* __restrict__ scores, float* __restrict__ fin_scores,
                                                         int32_t* __restrict__ fin_len, int64_t* __restrict__ fin_tokens,
                                                         int32_t* __restrict__ fin_count, int32_t* __restrict__ done) {
  constexpr int n_bm = NB;
  __shared__ float w_val[BEAM_MAX];
  __shared__ int w_idx[BEAM_MAX];
  __shared__ int64_t s_seq[BEAM_MAX * 64];
  __shared__ float red_v[BEAM_THREADS / 64];
  __shared__ int red_i[BEAM_THREADS / 64];
  __shared__ int red_t[BEAM_THREADS / 64];
  const int b = blockIdx.x;
  if (done[b]) return;
  const int tid = threadIdx.x;
  const int n_rows = (t == 1) ? 1 : n_bm;           // first step: only beam 0 holds <bos> (Beam.py:79-80)
  const float NEG = -3.0e38f;
  float lv[NB];
  int li[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) { lv[k] = NEG; li[k] = 0x7fffffff; }
  // per-row state first (one round trip for all beams), then one trip of loads for EVERY beam at a time
  bool ended[NB];
  float base[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    ended[r] = (r < n_rows) && (t > 1) && seqs[((int64_t)b * n_bm + r) * max_len + (t - 1)] == NACF_EOS;
    base[r] = (t > 1) ? scores[b * n_bm + r] : 0.f;
  }
  for (int v = tid; v < V; v += BEAM_THREADS) {
    float raw[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) raw[r] = (r < n_rows && !ended[r]) ? logp[((int64_t)b * n_bm + r) * ldl + v] : 0.f;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      // rows beyond n_rows (step 1) never win: NEG with the largest index loses every comparison against a real candidate
      float cv = (r < n_rows) ? (ended[r] ? -1e20f : raw[r] + base[r]) : NEG;      // beam_lk[i] = -1e20 (Beam.py:76-77)
      int ci = (r < n_rows) ? r * V + v : 0x7fffffff;
#pragma unroll
      for (int k = 0; k < NB; ++k) {         // descending value, ascending index on ties
        const bool better = cv > lv[k] || (cv == lv[k] && ci < li[k]);
        const float nv = better ? cv : lv[k];
        const int ni = better ? ci : li[k];
        cv = better ? lv[k] : cv;
        ci = better ? li[k] : ci;
        lv[k] = nv; li[k] = ni;
      }
    }
  }
  // n_bm rounds of "block argmax over every thread's current head"
  int head = 0;
  for (int k = 0; k < n_bm; ++k) {
    float hv = NEG;
    int hi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j == head) { hv = lv[j]; hi = li[j]; }
    if (head >= n_bm) { hv = NEG; hi = 0x7fffffff; }
    float bv = hv;
    int bi = hi, bt = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int ot = __shfl_xor(bt, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bt = ot; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; red_t[tid >> 6] = bt; }
    __syncthreads();
    bv = red_v[0]; bi = red_i[0]; bt = red_t[0];
    for (int w = 1; w < BEAM_THREADS / 64; ++w)
      if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; bt = red_t[w]; }
    if (tid == bt) head++;
    if (tid == 0) { w_val[k] = bv; w_idx[k] = bi; }
    __syncthreads();
  }
  // reorder the hypotheses by back-pointer and append the new word (prev_ks / next_ys, Beam.py:91-97)
  for (int e = tid; e < n_bm * max_len; e += BEAM_THREADS) {
    const int k = e / max_len, l = e % max_len;
    const int pk = w_idx[k] / V;
    s_seq[e] = (l == t) ? (int64_t)(w_idx[k] - pk * V) : seqs[((int64_t)b * n_bm + pk) * max_len + l];
  }
  __syncthreads();
  for (int e = tid; e < n_bm * max_len; e += BEAM_THREADS) seqs[(int64_t)b * n_bm * max_len + e] = s_seq[e];
  if (tid < n_bm) scores[b * n_bm + tid] = w_val[tid];
  __syncthreads();
  if (tid == 0) {
    int cnt = fin_count[b];
    bool is_done = false;
    for (int i = 0; i < n_bm && !is_done; ++i) {
      if (s_seq[i * max_len + t] == NACF_EOS) {                  // Beam.py:99-103
        if (cnt < want) {
          fin_scores[b * want + cnt] = w_val[i];
          fin_len[b * want + cnt] = t;
          for (int l = 0; l < max_len; ++l) fin_tokens[((int64_t)b * want + cnt) * max_len + l] = s_seq[i * max_len + l];
        }
        cnt++;
        if (cnt >= want) is_done = true;
      }
    }
    if (!is_done && t + 1 == max_len) {                          // Beam.py:116-121: len(next_ys) == max_len
      is_done = true;
      if (cnt == 0) {
        for (int i = 0; i < n_bm; ++i) {
          if (cnt < want) {
            fin_scores[b * want + cnt] = w_val[i];
            fin_len[b * want + cnt] = t;
            for (int l = 0; l < max_len; ++l) fin_tokens[((int64_t)b * want + cnt) * max_len + l] = s_seq[i * max_len + l];
          }
          cnt++;
          if (cnt >= want) break;
        }
      }
    }
    fin_count[b] = cnt < want ? cnt : want;
    if (is_done) done[b] = 1;
  }
}

__global__ void count_active_kernel(const int32_t* __restrict__ done, int B, int32_t* __restrict__ n_active) {
  __shared__ int red[256];
  int c = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) c += done[b] ? 0 : 1;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) n_active[0] = red[0];
}

inline int flat_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int nacf_beam_step(const float* logp, int64_t ldl, int B, int n_bm, int V, int t, int max_len, int want,
                   int64_t* seqs, float* scores, float* fin_scores, int32_t* fin_len, int64_t* fin_tokens,
                   int32_t* fin_count, int32_t* done, int32_t* n_active, nacf_stream_t stream) {
  NACF_CHECK(logp && seqs && scores && fin_scores && fin_len && fin_tokens && fin_count && done, NACF_EINVAL,
             "nacf_beam_step: null pointer");
  NACF_CHECK(B > 0 && V > 0 && n_bm >= 1 && n_bm <= BEAM_MAX && want >= 1 && want <= BEAM_MAX, NACF_EUNSUPPORTED,
             "nacf_beam_step: beam_size / topk must be in 1..%d", BEAM_MAX);
  NACF_CHECK(max_len >= 2 && max_len <= 64 && t >= 1 && t < max_len, NACF_EINVAL, "nacf_beam_step: bad step/max_len");
  NACF_CHECK((long)n_bm * V < 0x7fffffffL, NACF_EUNSUPPORTED, "nacf_beam_step: beam x vocab too large");
  hipStream_t s = as_hip(stream);
#define NACF_BEAM(NB)                                                                                                    \
  hipLaunchKernelGGL(beam_step_kernel<NB>, dim3(B), dim3(BEAM_THREADS), 0, s, logp, ldl, V, t, max_len, want, seqs, scores, \
                     fin_scores, fin_len, fin_tokens, fin_count, done)
  switch (n_bm) {
    case 1: NACF_BEAM(1); break;
    case 2: NACF_BEAM(2); break;
    case 3: NACF_BEAM(3); break;
    case 4: NACF_BEAM(4); break;
    case 5: NACF_BEAM(5); break;
    case 6: NACF_BEAM(6); break;
    case 7: NACF_BEAM(7); break;
    default: NACF_BEAM(8); break;
  }
#undef NACF_BEAM
  if (n_active) hipLaunchKernelGGL(count_active_kernel, dim3(1), dim3(256), 0, s, done, B, n_active);
  NACF_LAUNCH_CHECK("nacf_beam_step");
  return NACF_OK;
}

int nacf_token_replace(int64_t* tokens, int64_t n, int64_t from, int64_t to, nacf_stream_t stream) {
  NACF_CHECK(tokens && n > 0, NACF_EINVAL, "nacf_token_replace: bad argument");
  hipLaunchKernelGGL(token_replace_kernel, dim3(flat_grid(n)), dim3(256), 0, as_hip(stream), tokens, n, from, to);
  NACF_LAUNCH_CHECK("nacf_token_replace");
  return NACF_OK;
}
int nacf_teacher_probs(const float* label_logp, const int64_t* pad_tokens, float* out, int64_t n,
                       nacf_stream_t stream) {
  NACF_CHECK(label_logp && pad_tokens && out && n > 0, NACF_EINVAL, "nacf_teacher_probs: bad argument");
  hipLaunchKernelGGL(teacher_probs_kernel, dim3(flat_grid(n)), dim3(256), 0, as_hip(stream), label_logp, pad_tokens, out, n);
  NACF_LAUNCH_CHECK("nacf_teacher_probs");
  return NACF_OK;
}
int nacf_init_probs(const int64_t* pad_tokens, float* probs, int64_t n, nacf_stream_t stream) {
  NACF_CHECK(pad_tokens && probs && n > 0, NACF_EINVAL, "nacf_init_probs: bad argument");
  hipLaunchKernelGGL(init_probs_kernel, dim3(flat_grid(n)), dim3(256), 0, as_hip(stream), pad_tokens, probs, n);
  NACF_LAUNCH_CHECK("nacf_init_probs");
  return NACF_OK;
}
int nacf_apply_mask(int64_t* tokens, const uint8_t* mask, int64_t value, int64_t n, nacf_stream_t stream) {
  NACF_CHECK(tokens && mask && n > 0, NACF_EINVAL, "nacf_apply_mask: bad argument");
  hipLaunchKernelGGL(apply_mask_kernel, dim3(flat_grid(n)), dim3(256), 0, as_hip(stream), tokens, mask, value, n);
  NACF_LAUNCH_CHECK("nacf_apply_mask");
  return NACF_OK;
}
int nacf_mask_rank(const int64_t* tokens, int rows, int Lp, int32_t* rank, int32_t* counts, nacf_stream_t stream) {
  NACF_CHECK(tokens && rank && counts && rows > 0, NACF_EINVAL, "nacf_mask_rank: bad argument");
  NACF_CHECK(Lp > 0 && Lp <= 64, NACF_EUNSUPPORTED, "nacf_mask_rank: Lp must be in 1..64");
  hipLaunchKernelGGL(mask_rank_kernel, dim3(1), dim3(256), 0, as_hip(stream), tokens, rows, Lp, rank, counts);
  NACF_LAUNCH_CHECK("nacf_mask_rank");
  return NACF_OK;
}
int nacf_select_rank(const int32_t* rank, int cur, int q, int rows, int Lp, int64_t* tokens, uint8_t* mask_out,
                     nacf_stream_t stream) {
  NACF_CHECK(rank && tokens && mask_out && rows > 0 && Lp > 0 && q > 0, NACF_EINVAL, "nacf_select_rank: bad argument");
  const int64_t n = (int64_t)rows * Lp;
  hipLaunchKernelGGL(select_rank_kernel, dim3(flat_grid(n)), dim3(256), 0, as_hip(stream), rank, cur, q, n, tokens, mask_out);
  NACF_LAUNCH_CHECK("nacf_select_rank");
  return NACF_OK;
}
int nacf_easy_first_update(int64_t* tokens, float* probs, const int64_t* new_tokens, const float* new_probs, int q,
                           int rows, int Lp, nacf_stream_t stream) {
  NACF_CHECK(tokens && probs && new_tokens && new_probs && rows > 0 && q > 0, NACF_EINVAL,
             "nacf_easy_first_update: bad argument");
  NACF_CHECK(Lp > 0 && Lp <= 64, NACF_EUNSUPPORTED, "nacf_easy_first_update: Lp must be in 1..64");
  hipLaunchKernelGGL(easy_first_update_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_hip(stream), tokens, probs,
                     new_tokens, new_probs, q, rows, Lp);
  NACF_LAUNCH_CHECK("nacf_easy_first_update");
  return NACF_OK;
}

int nacf_length_beam(const float* pred_length, int B, int max_len, int lbs, int length_bias, int32_t* beam,
                     int32_t* beam_max, nacf_stream_t stream) {
  NACF_CHECK(pred_length && beam && beam_max, NACF_EINVAL, "nacf_length_beam: null pointer");
  NACF_CHECK(B > 0 && max_len > 4 && max_len <= 64 && lbs > 0 && lbs <= max_len, NACF_EINVAL,
             "nacf_length_beam: need 4 < max_len <= 64 and 0 < lbs <= max_len");
  hipLaunchKernelGGL(length_beam_kernel, dim3(1), dim3(256), 0, as_hip(stream), pred_length, B, max_len, lbs, length_bias,
                     beam, beam_max);
  NACF_LAUNCH_CHECK("nacf_length_beam");
  return NACF_OK;
}

int nacf_length_beam_gold(const int64_t* tgt_tokens, int B, int T, int max_len, int lbs, int32_t* beam, int32_t* beam_max,
                          nacf_stream_t stream) {
  NACF_CHECK(tgt_tokens && beam && beam_max, NACF_EINVAL, "nacf_length_beam_gold: null pointer");
  NACF_CHECK(B > 0 && T > 0 && max_len > 4 && max_len <= 64 && lbs > 0 && lbs <= max_len, NACF_EINVAL,
             "nacf_length_beam_gold: need T > 0, 4 < max_len <= 64 and 0 < lbs <= max_len");
  hipLaunchKernelGGL(length_beam_gold_kernel, dim3(1), dim3(256), 0, as_hip(stream), tgt_tokens, B, T, max_len, lbs, beam,
                     beam_max);
  NACF_LAUNCH_CHECK("nacf_length_beam_gold");
  return NACF_OK;
}

int nacf_canvas_init(const int32_t* beam, int rows, int Lp, int64_t* tokens, nacf_stream_t stream) {
  NACF_CHECK(beam && tokens && rows > 0 && Lp > 0, NACF_EINVAL, "nacf_canvas_init: bad argument");
  const int64_t total = (int64_t)rows * Lp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(canvas_init_kernel, dim3(blocks), dim3(256), 0, as_hip(stream), beam, rows, Lp, tokens);
  NACF_LAUNCH_CHECK("nacf_canvas_init");
  return NACF_OK;
}

int nacf_canvas_init_gold(const int32_t* beam, const int64_t* tgt_tokens, int T, int rows, int lbs, int Lp, int64_t* tokens,
                          nacf_stream_t stream) {
  NACF_CHECK(beam && tgt_tokens && tokens && rows > 0 && Lp > 0 && T > 0 && lbs > 0 && rows % lbs == 0, NACF_EINVAL,
             "nacf_canvas_init_gold: bad argument");
  const int64_t total = (int64_t)rows * Lp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(canvas_init_gold_kernel, dim3(blocks), dim3(256), 0, as_hip(stream), beam, tgt_tokens, T, rows, lbs, Lp,
                     tokens);
  NACF_LAUNCH_CHECK("nacf_canvas_init_gold");
  return NACF_OK;
}

int nacf_select_mask(const float* probs, const float* teacher_probs, const int64_t* pad_tokens,
                     const int32_t* num_mask_lut, int mode, int rows, int Lp, int64_t* tokens, uint8_t* mask_out,
                     nacf_stream_t stream) {
  NACF_CHECK(pad_tokens && tokens && mask_out && rows > 0, NACF_EINVAL, "nacf_select_mask: bad argument");
  NACF_CHECK(Lp > 0 && Lp <= 64, NACF_EUNSUPPORTED, "nacf_select_mask: Lp must be in 1..64");
  NACF_CHECK(mode >= 0 && mode <= 2, NACF_EINVAL, "nacf_select_mask: bad mode");
  NACF_CHECK(mode != 0 || (probs && num_mask_lut), NACF_EINVAL, "nacf_select_mask: mode 0 needs probs and lut");
  hipLaunchKernelGGL(select_mask_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_hip(stream), probs, teacher_probs,
                     pad_tokens, num_mask_lut, mode, rows, Lp, tokens, mask_out);
  NACF_LAUNCH_CHECK("nacf_select_mask");
  return NACF_OK;
}

int nacf_best_candidate(const int64_t* tokens, const float* probs, const float* teacher_probs, const int32_t* beam,
                        float alpha, int B, int lbs, int Lp, int64_t* out_tokens, int32_t* best_idx, float* cand_lprobs,
                        nacf_stream_t stream) {
  NACF_CHECK(tokens && probs && beam && out_tokens && B > 0 && lbs > 0, NACF_EINVAL, "nacf_best_candidate: bad argument");
  NACF_CHECK(Lp > 0 && Lp <= 64, NACF_EUNSUPPORTED, "nacf_best_candidate: Lp must be in 1..64");
  hipLaunchKernelGGL(best_candidate_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_hip(stream), tokens, probs, teacher_probs,
                     beam, alpha, B, lbs, Lp, out_tokens, best_idx, cand_lprobs);
  NACF_LAUNCH_CHECK("nacf_best_candidate");
  return NACF_OK;
}

}  // extern "C"
