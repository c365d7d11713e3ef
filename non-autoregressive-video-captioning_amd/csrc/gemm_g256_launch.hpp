// Host interface of the 256 x 256 x 64 eight-phase kernels (gemm_g256.hpp, nacf_gemm_g256.hip) towards nacf_gemm.hip.
#pragma once
#include <vector>
#include "common.hpp"

// ---- weight gradients of the bf16-matrix-core modes: dW[N][K] (+)= dZ[rows][N]^T X[rows][K] over the live rows
struct G256DwItem {
  const float* dZ; const float* X; float* dW; float* db; void* ws;
  size_t ws_bytes;
  int64_t lddz, ldx, lddw;
  int M, N, K;
  const int* rows;        // live-row list or nullptr
  const int* count;       // device count of live rows or nullptr (all M rows)
  float beta;
};
// what the caller's combine launch has to add up afterwards: dW = beta dW + sum of `splits` slabs, db = beta db + sum of
// `part_rows` rows of `part`
struct G256Reduce {
  float* slabs; float* dW; int64_t lddw;
  const float* part; float* db;
  int N, K, splits, part_rows;
  float beta;
};
bool g256_dw_enabled(int mode);                              // mode: NACF_GEMM_*
size_t g256_dw_workspace(int M, int N, int K);                 // bytes of `ws` an item needs on this path
// ONE grouped GEMM launch per 24 problems (ns = 1: throughput mode, 3: exact mode); appends the combines of the split problems
int g256_dw_group_launch(const G256DwItem* items, int n, int ns, std::vector<G256Reduce>& reduces, int* n_workgroups, hipStream_t s);
