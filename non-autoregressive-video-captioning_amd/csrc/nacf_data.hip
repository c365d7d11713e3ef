// Batch construction on the device (reference: dataloader.py).  The reference builds every
// sample in Python on the host (HDF5 read, numpy frame sampling, per-caption masking) with 0 workers; here the
// feature shards live in HBM (or arrive through a pinned staging buffer) and one launch per tensor family builds the
// whole batch:  frame selection + gather (HBM-bound), and the token / label tables of all captions (integer work).
#include <algorithm>

#include "common.hpp"

namespace {

// bound[i] = int(np.linspace(0, total, n+1)[i]) = int(i * (total / n)), last one = total   (dataloader.py:27)
__device__ __forceinline__ int seg_bound(int i, int total, int n) {
  if (i >= n) return total;
  const double step = (double)total / (double)n;
  return (int)((double)i * step);
}

__device__ __forceinline__ uint32_t rand_below(uint32_t r, uint32_t range) {   // [0, range)
  return (uint32_t)(((uint64_t)r * (uint64_t)range) >> 32);
}

// out[b, i, :] = src[video[b], frame(b, i), :]
//   mode 0 'equally_sampling': middle of segment i;  mode 1 'segment_random': uniform in segment i;  mode 2 'all_random'
//   (dataloader.py:24-37);  a clip shorter than n_frames is stretched: round-half-even(i * (S-1) / (n-1)) (:20-21,305)
__device__ __forceinline__ int pick_frame(int b, int i, int S, int n_frames, int mode, uint32_t salt,
                                          const uint64_t* __restrict__ rng_state) {
  if (S < n_frames) return (n_frames > 1) ? (int)rint((double)(i * (S - 1)) / (double)(n_frames - 1)) : 0;
  if (mode == 2) {
    // 'all_random' (dataloader.py:25-26,37): n_frames distinct frames of the clip, ascending.  Selection sampling: frame t is
    // taken with probability (still needed) / (still left), which draws every n-subset with equal probability and yields it
    // in ascending order -- so the i-th frame taken is sorted(random.sample(range(S), n))[i] in distribution; one Philox
    // word per frame, the same words in every (b, i) block of a clip
    DropRng rng;
    rng.init(rng_state);
    int taken = 0;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    for (int t = 0; t < S; ++t) {
      if ((t & 3) == 0) r = philox4x32_10(make_uint4((uint32_t)b, salt, rng.step_lo ^ (uint32_t)(t >> 2), rng.step_hi ^ 0xA11C0FFEu), rng.key);
      const uint32_t w = (t & 3) == 0 ? r.x : (t & 3) == 1 ? r.y : (t & 3) == 2 ? r.z : r.w;
      if ((int)rand_below(w, (uint32_t)(S - t)) < n_frames - taken) {
        if (taken == i) return t;
        ++taken;
      }
    }
    return S - 1;      // not reached: the last frames are taken with probability 1 once as many are needed as are left
  }
  const int lo = seg_bound(i, S, n_frames), hi = seg_bound(i + 1, S, n_frames);
  if (mode != 1) return (lo + hi) / 2;
  if (hi <= lo + 1) return lo;
  DropRng rng;
  rng.init(rng_state);
  const uint64_t e = (uint64_t)b * (uint64_t)n_frames + (uint64_t)i;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)e, salt ^ (uint32_t)(e >> 32), rng.step_lo, rng.step_hi), rng.key);
  return lo + (int)rand_below(r.x, (uint32_t)(hi - lo));
}

__global__ __launch_bounds__(256) void sample_frames_kernel(const float* __restrict__ src, const int* __restrict__ video,
                                                            const int* __restrict__ src_len, int T, int D, int n_frames,
                                                            int mode, uint32_t salt, const uint64_t* __restrict__ rng_state,
                                                            float* __restrict__ out, int* __restrict__ ids_out) {
  const int b = blockIdx.x, i = blockIdx.y;
  const int S = src_len ? min(src_len[video ? video[b] : b], T) : T;
  const int f = pick_frame(b, i, S, n_frames, mode, salt, rng_state);
  if (threadIdx.x == 0 && ids_out) ids_out[b * n_frames + i] = f;
  const int64_t v = video ? video[b] : b;
  const float* p = src + (v * T + f) * (int64_t)D;
  float* q = out + ((int64_t)b * n_frames + i) * (int64_t)D;
  if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    for (int d = threadIdx.x * 4; d < D; d += 1024) *reinterpret_cast<f32x4*>(q + d) = *reinterpret_cast<const f32x4*>(p + d);
  } else {
    for (int d = threadIdx.x; d < D; d += 256) q[d] = p[d];
  }
}

// The same gather when `src` is PINNED HOST memory read over PCIe (zero-copy): a few persistent workgroups instead of
// one per row -- the link needs ~100 KB in flight, not thousands of resident waves that would take compute units
// away from the training kernels running next to it -- and each thread keeps ROWS_IN_FLIGHT rows' loads outstanding
// before it stores any of them.
constexpr int ROWS_IN_FLIGHT = 4;
__global__ __launch_bounds__(256) void sample_frames_pcie_kernel(const float* __restrict__ src, const int* __restrict__ video,
                                                                 const int* __restrict__ src_len, int B, int T, int D,
                                                                 int n_frames, int mode, uint32_t salt,
                                                                 const uint64_t* __restrict__ rng_state,
                                                                 float* __restrict__ out, int* __restrict__ ids_out) {
  const int total = B * n_frames;
  const int chunks = D / 1024;                      // f32x4 per thread per row (host guarantees D % 1024 == 0, <= 4 chunks)
  for (int r0 = blockIdx.x * ROWS_IN_FLIGHT; r0 < total; r0 += gridDim.x * ROWS_IN_FLIGHT) {
    f32x4 buf[ROWS_IN_FLIGHT][4];
#pragma unroll
    for (int k = 0; k < ROWS_IN_FLIGHT; ++k) {
      const int r = r0 + k;
      if (r < total) {
        const int b = r / n_frames, i = r % n_frames;
        const int64_t v = video ? video[b] : b;
        const int S = src_len ? min(src_len[v], T) : T;
        const int f = pick_frame(b, i, S, n_frames, mode, salt, rng_state);
        if (threadIdx.x == 0 && ids_out) ids_out[r] = f;
        const float* p = src + (v * T + f) * (int64_t)D;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < chunks) buf[k][c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c * 1024 + threadIdx.x * 4));
      }
    }
#pragma unroll
    for (int k = 0; k < ROWS_IN_FLIGHT; ++k) {
      const int r = r0 + k;
      if (r < total) {
        float* q = out + (int64_t)r * D;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < chunks) *reinterpret_cast<f32x4*>(q + c * 1024 + threadIdx.x * 4) = buf[k][c];
      }
    }
  }
}

constexpr int MAX_SENT = 256;   // words per caption the masking sampler can permute (corpora cap captions far below)

// One thread per caption: source / target tables of dataloader.py:317-425.
//   caps[b, 0..len) = <bos> w1 .. wn <eos>;  tags alike (may be NULL when visual_word == 0)
__global__ void build_targets_kernel(const int* __restrict__ caps, int ld, const int* __restrict__ cap_len,
                                     const int* __restrict__ tags, const uint8_t* __restrict__ tag_demanded,
                                     const uint8_t* __restrict__ word_is_be, int B, int max_len, int narformer,
                                     int visual_word, int train, double beta_low, double beta_high, uint32_t salt,
                                     const uint64_t* __restrict__ rng_state, int64_t* __restrict__ tokens,
                                     int64_t* __restrict__ labels, int64_t* __restrict__ tokens_1,
                                     int64_t* __restrict__ labels_1) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int* c = caps + (int64_t)b * ld;
  const int len = cap_len[b];          // including <bos>, <eos>
  const int n = len - 2;               // words
  int64_t* tok = tokens + (int64_t)b * max_len;
  int64_t* lab = labels + (int64_t)b * max_len;
  if (narformer) {
    // ---- masked LM pair (:346-380)
    for (int i = 0; i < max_len; ++i) {
      const bool in = i < n;
      if (train) { tok[i] = in ? c[i + 1] : NACF_PAD; lab[i] = NACF_PAD; }
      else { tok[i] = in ? (c[i + 1] != NACF_PAD ? NACF_MASK : NACF_PAD) : NACF_PAD; lab[i] = in ? c[i + 1] : NACF_PAD; }
    }
    if (train && n > 1) {
      int low = (int)((double)n * beta_low), high = (int)((double)n * beta_high);
      low = low < 1 ? 1 : low;
      high = high < 1 ? 1 : high;
      if (high == low) high += 1;
      DropRng rng;
      rng.init(rng_state);
      uint32_t ctr = 0;
      uint4 r = philox4x32_10(make_uint4((uint32_t)b, salt, rng.step_lo, rng.step_hi ^ (ctr++ << 16)), rng.key);
      int have = 3;                                     // r.x is spent on the count below
      const int k = low + (int)rand_below(r.x, (uint32_t)(high - low));      // number of masked slots in [low, high)
      // k positions without replacement: partial Fisher-Yates over 0..n-1 (positions beyond max_len are drawn too,
      // exactly like the reference, which masks the full sentence and truncates afterwards)
      uint8_t perm[MAX_SENT];
      const int nn = n < MAX_SENT ? n : MAX_SENT;
      for (int i = 0; i < nn; ++i) perm[i] = (uint8_t)i;
      for (int j = 0; j < k && j < nn; ++j) {
        if (have == 0) { r = philox4x32_10(make_uint4((uint32_t)b, salt, rng.step_lo, rng.step_hi ^ (ctr++ << 16)), rng.key); have = 4; }
        const uint32_t rv = have == 4 ? r.x : (have == 3 ? r.y : (have == 2 ? r.z : r.w));
        --have;
        const int pick = j + (int)rand_below(rv, (uint32_t)(nn - j));
        const uint8_t t = perm[pick]; perm[pick] = perm[j]; perm[j] = t;
        const int pos = t;
        if (pos < max_len) { lab[pos] = c[pos + 1]; tok[pos] = NACF_MASK; }
      }
    }
  } else {
    // ---- autoregressive pair: padding(target, add_eos=True) for both (:333-337)
    for (int i = 0; i < max_len; ++i) {
      int64_t v = i < len ? c[i] : NACF_PAD;
      if (len > max_len && i == max_len - 1) v = NACF_EOS;
      tok[i] = v;
      lab[i] = v;
    }
  }
  if (visual_word && train) {
    // ---- visual-word pair (:382-425)
    int64_t* t1 = tokens_1 + (int64_t)b * max_len;
    int64_t* l1 = labels_1 + (int64_t)b * max_len;
    const int* tg = tags + (int64_t)b * ld;
    if (narformer) {
      for (int i = 0; i < max_len; ++i) {
        const bool in = i < n;
        t1[i] = in ? NACF_VIS : NACF_PAD;
        int64_t v = NACF_PAD;
        if (in) v = (tag_demanded[tg[i + 1]] && !word_is_be[c[i + 1]]) ? c[i + 1] : NACF_MASK;
        l1[i] = v;
      }
    } else {
      for (int i = 0; i < max_len; ++i) {
        int64_t s = i < len ? NACF_VIS : NACF_PAD;
        if (len > max_len && i == max_len - 1) s = NACF_EOS;
        t1[i] = s;
        int64_t v = NACF_PAD;                       // [<bos>] + word-or-<mask> * n + [<eos>], padded with add_eos
        if (i == 0) v = c[0];
        else if (i <= n) v = (tag_demanded[tg[i]] && !word_is_be[c[i]]) ? c[i] : NACF_MASK;
        else if (i == n + 1) v = NACF_EOS;
        if (len > max_len && i == max_len - 1) v = NACF_EOS;
        l1[i] = v;
      }
    }
  }
}

// Whole clips out of PINNED HOST memory by kernel-issued PCIe reads (the pinned allocation is mapped into the device's address
// space): dst[j] = src[rows[j]].  A few workgroups keep ZC_UNROLL x 16 bytes per thread in flight (24 workgroups: 1.5 MB of reads
// outstanding) while the training step owns the rest of the chip.  One DMA per clip
// (nacf_gather_clips_h2d) tops out at ~26 GB/s for the 491 KB clips of a 60 x 2048 batch; this form is what the pinned-host
// placement of the loader uses when the whole clip is needed (data/loader.py).
typedef uint32_t zc_u32x4 __attribute__((ext_vector_type(4)));
#ifndef ZC_UNROLL
#define ZC_UNROLL 16
#endif
__global__ __launch_bounds__(256) void gather_clips_zc_kernel(zc_u32x4* __restrict__ dst, const zc_u32x4* __restrict__ src,
                                                              const int32_t* __restrict__ rows, int n, int64_t chunks) {
  constexpr int U = ZC_UNROLL;
  const int64_t per_clip = (chunks + 256 * U - 1) / (256 * U);
  for (int64_t w = blockIdx.x; w < (int64_t)n * per_clip; w += gridDim.x) {
    const int j = (int)(w / per_clip);
    const int64_t c0 = (w - (int64_t)j * per_clip) * (256 * U) + threadIdx.x;
    const zc_u32x4* s = src + (int64_t)rows[j] * chunks;
    zc_u32x4* d = dst + (int64_t)j * chunks;
    zc_u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (c0 + u * 256 < chunks) v[u] = __builtin_nontemporal_load(s + c0 + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (c0 + u * 256 < chunks) d[c0 + u * 256] = v[u];
  }
}

}  // namespace

extern "C" {

int nacf_sample_frames(const float* src, const int32_t* video, const int32_t* src_len, int B, int T, int D, int n_frames,
                       int mode, uint32_t salt, const uint64_t* rng_state, float* out, int32_t* frame_ids,
                       nacf_stream_t stream) {
  NACF_CHECK(src && out && B > 0 && T > 0 && D > 0 && n_frames > 0, NACF_EINVAL, "nacf_sample_frames: bad argument");
  NACF_CHECK(mode >= 0 && mode <= 2, NACF_EINVAL,
             "nacf_sample_frames: mode must be 0 (equally_sampling), 1 (segment_random) or 2 (all_random)");
  NACF_CHECK(!(mode == 1 && !rng_state), NACF_EINVAL, "nacf_sample_frames: segment_random needs rng_state");
  hipPointerAttribute_t attr;
  const bool host_src = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();          // an unregistered pointer is an error for the query only
  if (host_src && D % 1024 == 0 && D <= 4096 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int total = B * n_frames;
    const int grid = std::min(96, cdiv(total, ROWS_IN_FLIGHT));
    hipLaunchKernelGGL(sample_frames_pcie_kernel, dim3(grid), dim3(256), 0, as_hip(stream), src, video, src_len, B, T, D,
                       n_frames, mode, salt, rng_state, out, frame_ids);
    NACF_LAUNCH_CHECK("nacf_sample_frames(pcie)");
    return NACF_OK;
  }
  hipLaunchKernelGGL(sample_frames_kernel, dim3(B, n_frames), dim3(256), 0, as_hip(stream), src, video, src_len, T, D,
                     n_frames, mode, salt, rng_state, out, frame_ids);
  NACF_LAUNCH_CHECK("nacf_sample_frames");
  return NACF_OK;
}

int nacf_gather_clips_h2d(void* dst, const void* src_host, const int32_t* rows, int n, size_t clip_bytes,
                          nacf_stream_t stream) {
  NACF_CHECK(dst && src_host && rows && n >= 0 && clip_bytes > 0, NACF_EINVAL, "nacf_gather_clips_h2d: bad argument");
  hipStream_t s = as_hip(stream);
  for (int j = 0; j < n; ++j) {
    NACF_CHECK(rows[j] >= 0, NACF_EINVAL, "nacf_gather_clips_h2d: negative row %d", rows[j]);
    const hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + (size_t)j * clip_bytes,
                                        static_cast<const char*>(src_host) + (size_t)rows[j] * clip_bytes, clip_bytes,
                                        hipMemcpyHostToDevice, s);
    NACF_CHECK(e == hipSuccess, NACF_ELAUNCH, "nacf_gather_clips_h2d: %s", hipGetErrorString(e));
  }
  return NACF_OK;
}

int nacf_gather_clips_zc(void* dst, const void* src_host, const int32_t* rows_dev, int n, size_t clip_bytes, int workgroups,
                         nacf_stream_t stream) {
  NACF_CHECK(dst && src_host && rows_dev && n >= 0 && clip_bytes > 0 && clip_bytes % 16 == 0, NACF_EINVAL, "nacf_gather_clips_zc: bad argument");
  NACF_CHECK((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src_host) & 15) == 0, NACF_EINVAL,
             "nacf_gather_clips_zc: 16-byte aligned buffers");
  if (n == 0) return NACF_OK;
  if (workgroups <= 0) workgroups = 24;
  hipLaunchKernelGGL(gather_clips_zc_kernel, dim3(workgroups), dim3(256), 0, as_hip(stream), static_cast<zc_u32x4*>(dst),
                     static_cast<const zc_u32x4*>(src_host), rows_dev, n, (int64_t)(clip_bytes / 16));
  NACF_LAUNCH_CHECK("nacf_gather_clips_zc");
  return NACF_OK;
}

int nacf_build_targets(const int32_t* caps, int ld_caps, const int32_t* cap_len, const int32_t* pos_tags,
                       const uint8_t* tag_demanded, const uint8_t* word_is_be, int B, int max_len, int narformer,
                       int visual_word, int train, double beta_low, double beta_high, uint32_t salt,
                       const uint64_t* rng_state, int64_t* tokens, int64_t* labels, int64_t* tokens_1, int64_t* labels_1,
                       nacf_stream_t stream) {
  NACF_CHECK(caps && cap_len && tokens && labels && B > 0 && max_len > 0 && ld_caps > 0, NACF_EINVAL,
             "nacf_build_targets: bad argument");
  NACF_CHECK(!(visual_word && train && !(pos_tags && tag_demanded && word_is_be && tokens_1 && labels_1)), NACF_EINVAL,
             "nacf_build_targets: visual-word targets need pos_tags, both look-up tables and the *_1 outputs");
  NACF_CHECK(!(narformer && train && !rng_state), NACF_EINVAL, "nacf_build_targets: training masks need rng_state");
  hipLaunchKernelGGL(build_targets_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_hip(stream), caps, ld_caps, cap_len, pos_tags,
                     tag_demanded, word_is_be, B, max_len, narformer, visual_word, train, beta_low, beta_high, salt,
                     rng_state, tokens, labels, tokens_1, labels_1);
  NACF_LAUNCH_CHECK("nacf_build_targets");
  return NACF_OK;
}

}  // extern "C"
