// Small-state fp64 GEMM on the matrix cores for the Metropolis-Hastings steps:  C = A0 B0 (+ A1 B1) (+ v 1'),
// A shared d x d (column-major), B the d x C state of all chains (column-major = chain-major rows), d ~ 500, C ~ 512.
//
// Why an own kernel: the products of a step are tiny by GEMM standards (500 x 500 x 512 = 0.26 Gflop); a library tile
// choice made for throughput covers 16-32 of the 256 CUs with them, and the step is a chain of
// such products.  Here the output is cut into 64 x 16 tiles -- 8 x 32 = 256 workgroups at cfg4, 768 for the triple
// product -- each workgroup 4 waves, each wave ONE v_mfma_f64_16x16x4_f64 accumulator (16 rows x 16 chains) walked over
// the whole contraction (64 x 32 tiles halve the workgroups and measured slower: 20.6 against 15.7 us per product); A and the state are L2-resident (2 MB each), so the small tile's extra operand traffic is
// L2 -> LDS traffic, not HBM.  Two operand pairs in one launch fuse x' = A1p x + L^-T z into one pass (no second launch,
// no read-modify-write of the output); an optional vector is added to every column in the epilogue (the constant part
// of the proposal mean); `tri` skips the slabs of an upper-triangular A that are identically zero (the L' products).
//   As[k][i] (row stride 64 + 16): lane l reads A[i0 + l%16][k + l/16]: the two rows of a half-wave in different bank halves
//   Bs[j][k] (row stride BK + 2):  lane l reads B[k + l/16][j0 + l%16]: 16 columns x 2 rows tile all 64 banks exactly once
#include "omc_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GM_TM 64
#define GM_TN 16
#define GM_BK 32
#define GM_LDA (GM_TM + 16)
#define GM_LDB (GM_BK + 2)
#define GM_NACC 4

struct GemmPair {
  const double* A; int64_t lda;
  const double* B; int64_t ldb;
  int K;
};

typedef double double2_t __attribute__((ext_vector_type(2)));

// VEC: operands fetched 16 bytes per lane (M, the K's and the leading dimensions even, bases 16-byte aligned).
// KS: the contraction cut KS ways INSIDE the workgroup -- KS groups of four waves, each with its own LDS slabs and its
// own quarter of every operand pair, partial tiles added in group order through LDS at the end (deterministic).  What
// bounds these small products is how many operand bytes ONE resident workgroup per CU keeps in flight (DESIGN.md
// section 5.3); KS = 4 puts four times as many loads on the wire without more workgroups or a second kernel.
template <bool VEC, int KS>
__global__ void __launch_bounds__(256 * KS) k_dgemm_64x16(int M, int N, GemmPair p0, GemmPair p1, int tri, const double* __restrict__ addv,
                                                          double* __restrict__ Cout, int64_t ldc, const int* __restrict__ colmask) {
  __shared__ double As_all[KS][GM_BK][GM_LDA];
  __shared__ double Bs_all[KS][GM_TN][GM_LDB];
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  double(*As)[GM_LDA] = As_all[grp];
  double(*Bs)[GM_LDB] = Bs_all[grp];
  const int i0 = blockIdx.x * GM_TM, j0 = blockIdx.y * GM_TN;
  double4_t acc[GM_NACC];
#pragma unroll
  for (int t = 0; t < GM_NACC; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  double ra[GM_BK / 4], rb[GM_BK / 16];
  for (int pass = 0; pass < 2; ++pass) {
    const GemmPair P = pass ? p1 : p0;
    if (P.K <= 0) continue;  // uniform over the workgroup: the barriers below stay matched
    // upper-triangular A (A(i,k) = 0 for k < i): nothing to add before the tile's first row
    const int kbeg = tri ? (i0 / GM_BK) * GM_BK : 0;
    // this group's share of [kbeg, K): whole slabs, the same count for every group (idle slabs load zeros)
    const int nslab = (P.K - kbeg + GM_BK - 1) / GM_BK;
    const int per = (nslab + KS - 1) / KS;
    const int k_lo = kbeg + grp * per * GM_BK;
    const int k_hi = (k_lo + per * GM_BK < P.K) ? k_lo + per * GM_BK : P.K;  // loads beyond k_hi give zeros
    auto fetch = [&](int k0) {
      if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < GM_BK / 8; ++q) {  // A slab as pairs: pair e = q*256 + tid: k = e / 32, i = 2 (e % 32)
          const int e = q * 256 + tid, k = k0 + (e >> 5), i = i0 + 2 * (e & 31);
          double2_t v = {0.0, 0.0};
          if (k < k_hi && i < M) v = *reinterpret_cast<const double2_t*>(P.A + (int64_t)k * P.lda + i);
          ra[2 * q] = v.x; ra[2 * q + 1] = v.y;
        }
        {  // B slab as pairs: j = tid / 16, k = 2 (tid % 16)
          const int j = j0 + (tid >> 4), k = k0 + 2 * (tid & 15);
          double2_t v = {0.0, 0.0};
          if (k < k_hi && j < N) v = *reinterpret_cast<const double2_t*>(P.B + (int64_t)j * P.ldb + k);
          rb[0] = v.x; rb[1] = v.y;
        }
      } else {
#pragma unroll
        for (int q = 0; q < GM_BK / 4; ++q) {  // A slab: BK x 64, element e = q*256 + tid: k = e / 64, i = e % 64 (contiguous in i)
          const int e = q * 256 + tid, k = k0 + (e >> 6), i = i0 + (e & 63);
          ra[q] = (k < k_hi && i < M) ? P.A[(int64_t)k * P.lda + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GM_BK / 16; ++q) {  // B slab: BK x 16, element e: j = e / BK, k = e % BK (contiguous in k)
          const int e = q * 256 + tid, j = j0 + e / GM_BK, k = k0 + e % GM_BK;
          rb[q] = (k < k_hi && j < N) ? P.B[(int64_t)j * P.ldb + k] : 0.0;
        }
      }
    };
    if (per > 0) fetch(k_lo);
    for (int sl = 0; sl < per; ++sl) {
      const int k0 = k_lo + sl * GM_BK;
      __syncthreads();
      if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < GM_BK / 8; ++q) {
          const int e = q * 256 + tid;
          *reinterpret_cast<double2_t*>(&As[e >> 5][2 * (e & 31)]) = double2_t{ra[2 * q], ra[2 * q + 1]};
        }
        *reinterpret_cast<double2_t*>(&Bs[tid >> 4][2 * (tid & 15)]) = double2_t{rb[0], rb[1]};
      } else {
#pragma unroll
        for (int q = 0; q < GM_BK / 4; ++q) {
          const int e = q * 256 + tid;
          As[e >> 6][e & 63] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < GM_BK / 16; ++q) {
          const int e = q * 256 + tid;
          Bs[e / GM_BK][e % GM_BK] = rb[q];
        }
      }
      __syncthreads();
      if (sl + 1 < per) fetch(k0 + GM_BK);  // in flight under the multiplications
#pragma unroll
      for (int kk = 0; kk < GM_BK; kk += 4) {
        const int kr = kk + (lane >> 4), cl = lane & 15;
        acc[(kk / 4) % GM_NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(As[kr][wave * 16 + cl], Bs[cl][kr], acc[(kk / 4) % GM_NACC], 0, 0, 0);
      }
    }
  }
  double4_t v = acc[0];
#pragma unroll
  for (int t = 1; t < GM_NACC; ++t) v += acc[t];
  if constexpr (KS > 1) {  // partial tiles of the groups through LDS (the slabs are dead), added in group order
    __syncthreads();
    double* red = &As_all[0][0][0];  // KS x 256 x 4 doubles <= KS * GM_BK * GM_LDA
    if (grp > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((grp * 256 + tid) << 2) + r] = v[r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += red[((g * 256 + tid) << 2) + r];
  }
  // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
  const int j = j0 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + wave * 16 + (lane >> 4) + 4 * r;
    if (i < M && j < N && (!colmask || colmask[j])) Cout[(int64_t)j * ldc + i] = v[r] + (addv ? addv[i] : 0.0);
  }
}

// ---- the same product for MANY columns (the states of a block of steps: N = 32 C, omc_mala_run_white) ---------------------
// With thousands of columns the chip is full without small tiles, and the 64 x 16 tile's cost shows: every 16 columns re-read
// their 64 x K strip of A from L2 into LDS.  Here a workgroup makes a 64 x 64 tile -- each of its four waves 16 rows x 64
// columns = four accumulators fed by ONE A fragment per contraction step -- so A crosses L2 -> LDS a quarter as often and a wave
// issues four independent MFMAs per LDS round.  Upper-triangular A: the slabs in front of the tile's first row are skipped.
#define GW_TN 64
template <bool VEC>
__global__ void __launch_bounds__(256) k_dgemm_64x64(int M, int N, GemmPair P, int tri, const double* __restrict__ addv,
                                                     double* __restrict__ Cout, int64_t ldc) {
  __shared__ double As[1][GM_BK][GM_LDA];
  __shared__ double Bs[1][GW_TN][GM_LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * GM_TM, j0 = blockIdx.y * GW_TN;
  double4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int kbeg = tri ? (i0 / GM_BK) * GM_BK : 0;
  const int nslab = (P.K - kbeg + GM_BK - 1) / GM_BK;
  double ra[GM_BK / 4], rb[GM_BK / 4];
  auto fetch = [&](int k0) {
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < GM_BK / 8; ++q) {  // A slab as pairs: pair e = q*256 + tid: k = e / 32, i = 2 (e % 32)
        const int e = q * 256 + tid, k = k0 + (e >> 5), i = i0 + 2 * (e & 31);
        double2_t v = {0.0, 0.0};
        if (k < P.K && i < M) v = *reinterpret_cast<const double2_t*>(P.A + (int64_t)k * P.lda + i);
        ra[2 * q] = v.x; ra[2 * q + 1] = v.y;
      }
#pragma unroll
      for (int q = 0; q < GM_BK / 8; ++q) {  // B slab as pairs: pair e: j = e / 16, k = 2 (e % 16)
        const int e = q * 256 + tid, j = j0 + (e >> 4), k = k0 + 2 * (e & 15);
        double2_t v = {0.0, 0.0};
        if (k < P.K && j < N) v = *reinterpret_cast<const double2_t*>(P.B + (int64_t)j * P.ldb + k);
        rb[2 * q] = v.x; rb[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < GM_BK / 4; ++q) {
        const int e = q * 256 + tid, k = k0 + (e >> 6), i = i0 + (e & 63);
        ra[q] = (k < P.K && i < M) ? P.A[(int64_t)k * P.lda + i] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < GM_BK / 4; ++q) {
        const int e = q * 256 + tid, j = j0 + e / GM_BK, k = k0 + e % GM_BK;
        rb[q] = (k < P.K && j < N) ? P.B[(int64_t)j * P.ldb + k] : 0.0;
      }
    }
  };
  auto park = [&](int b) {
    if constexpr (VEC) {
#pragma unroll
      for (int q = 0; q < GM_BK / 8; ++q) {
        const int e = q * 256 + tid;
        *reinterpret_cast<double2_t*>(&As[b][e >> 5][2 * (e & 31)]) = double2_t{ra[2 * q], ra[2 * q + 1]};
        *reinterpret_cast<double2_t*>(&Bs[b][e >> 4][2 * (e & 15)]) = double2_t{rb[2 * q], rb[2 * q + 1]};
      }
    } else {
#pragma unroll
      for (int q = 0; q < GM_BK / 4; ++q) {
        const int e = q * 256 + tid;
        As[b][e >> 6][e & 63] = ra[q];
        Bs[b][e / GM_BK][e % GM_BK] = rb[q];
      }
    }
  };
  // One LDS buffer per operand (the next slab waits in registers): 37 KB per workgroup, so FOUR workgroups share a CU and their
  // loads, barriers and matrix-core phases interleave -- with two buffers (75 KB, two workgroups per CU) the counters showed the
  // waves waiting half their cycles on memory and barriers with the matrix cores 31 % busy.
  if (nslab > 0) fetch(kbeg);
  for (int sl = 0; sl < nslab; ++sl) {
    __syncthreads();                                             // everyone is done reading the previous slab
    park(0);
    __syncthreads();
    if (sl + 1 < nslab) fetch(kbeg + (sl + 1) * GM_BK);          // in flight under the multiplications
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 4) {
      const int kr = kk + (lane >> 4), cl = lane & 15;
      const double af = As[0][kr][wave * 16 + cl];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, Bs[0][16 * t + cl][kr], acc[t], 0, 0, 0);
    }
  }
  // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + 16 * t + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 16 + (lane >> 4) + 4 * r;
      if (i < M && j < N) Cout[(int64_t)j * ldc + i] = acc[t][r] + (addv ? addv[i] : 0.0);
    }
  }
}

omc_status omc_dgemm_wide(omc_ctx* ctx, int M, int N, const double* A, int64_t lda, const double* B, int64_t ldb, int K, int tri,
                          const double* addv, double* Cout, int64_t ldc) {
  GemmPair p{A, lda, B, ldb, K};
  const dim3 grid((unsigned)((M + GM_TM - 1) / GM_TM), (unsigned)((N + GW_TN - 1) / GW_TN));
  const bool vec = (M & 1) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15u) == 0 && (lda & 1) == 0 && (ldb & 1) == 0 && (K & 1) == 0;
  if (vec) hipLaunchKernelGGL((k_dgemm_64x64<true>), grid, dim3(256), 0, ctx->stream, M, N, p, tri, addv, Cout, ldc);
  else hipLaunchKernelGGL((k_dgemm_64x64<false>), grid, dim3(256), 0, ctx->stream, M, N, p, tri, addv, Cout, ldc);
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}

// C[M x N] = A0[M x K0] B0[K0 x N] (+ A1 B1) (+ addv per column); everything column-major, device pointers;
// colmask (N ints or NULL): only the columns with a non-zero entry are written
omc_status omc_dgemm_small(omc_ctx* ctx, int M, int N, const double* A0, int64_t lda0, const double* B0, int64_t ldb0, int K0,
                           const double* A1, int64_t lda1, const double* B1, int64_t ldb1, int K1, int tri, const double* addv,
                           double* Cout, int64_t ldc, const int* colmask, int ksplit) {
  GemmPair p0{A0, lda0, B0, ldb0, K0}, p1{A1, lda1, B1, ldb1, A1 ? K1 : 0};
  const dim3 grid((unsigned)((M + GM_TM - 1) / GM_TM), (unsigned)((N + GM_TN - 1) / GM_TN));
  auto even16 = [](const double* p, int64_t ld, int K) { return !p || (((uintptr_t)p & 15u) == 0 && (ld & 1) == 0 && (K & 1) == 0); };
  const bool vec = (M & 1) == 0 && even16(A0, lda0, K0) && even16(B0, ldb0, K0) && even16(A1, lda1, K1) && even16(B1, ldb1, K1);
  const int ks = ksplit > 0 ? ksplit : ctx->mh_gemm_ksplit;  // (a product with thousands of columns fills the chip without the in-workgroup split)
#define OMC_GEMM_LAUNCH(V, K) hipLaunchKernelGGL((k_dgemm_64x16<V, K>), grid, dim3(256 * K), 0, ctx->stream, M, N, p0, p1, tri, addv, Cout, ldc, colmask)
  if (vec) {
    if (ks >= 4) OMC_GEMM_LAUNCH(true, 4); else if (ks == 2) OMC_GEMM_LAUNCH(true, 2); else OMC_GEMM_LAUNCH(true, 1);
  } else {
    if (ks >= 4) OMC_GEMM_LAUNCH(false, 4); else if (ks == 2) OMC_GEMM_LAUNCH(false, 2); else OMC_GEMM_LAUNCH(false, 1);
  }
#undef OMC_GEMM_LAUNCH
  OMC_HIP_CHECK(hipGetLastError());
  return OMC_OK;
}
