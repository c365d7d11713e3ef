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
      for (int i = 0; i < max_len; ++i) {
        if ((taken >> i) & 1ull) continue;
        const float v = p[i];
        if (bi < 0 || v > best) { best = v; bi = i; }
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

__global__ void canvas_init_kernel(const int32_t* __restrict__ beam, int rows, int Lp, int64_t* __restrict__ tokens) {
  const int64_t total = (int64_t)rows * Lp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Lp), l = (int)(e % Lp);
    tokens[e] = l < beam[r] ? NACF_MASK : NACF_PAD;
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

}  // namespace

extern "C" {

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

int nacf_canvas_init(const int32_t* beam, int rows, int Lp, int64_t* tokens, nacf_stream_t stream) {
  NACF_CHECK(beam && tokens && rows > 0 && Lp > 0, NACF_EINVAL, "nacf_canvas_init: bad argument");
  const int64_t total = (int64_t)rows * Lp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(canvas_init_kernel, dim3(blocks), dim3(256), 0, as_hip(stream), beam, rows, Lp, tokens);
  NACF_LAUNCH_CHECK("nacf_canvas_init");
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
