// The diagonal block of the blocked batched Cholesky (omc_dense.hip, potrf_blocked_part): factor and inverse of a 64 x 64 block
// by ONE wave per chain, the block's rows in registers.  A translation unit of its own, free of the library's headers: the 64
// fully unrolled column steps take minutes to compile and change with nothing else.
//   k_chol_diag   lane r keeps row r of the block in registers; column k is scaled and the trailing columns updated with the pivot
//                 column's entries broadcast by v_readlane -- no LDS, no barrier on the column-to-column path; then L^-1 by
//                 forward substitution (lane c: column c), L read as LDS broadcasts.  Same operations in the same order per entry
//                 as k_chol_panel's factor: bit-identical L (gmrf.py:481: np.linalg.cholesky of the block, to rounding).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CH_NB 64

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int K>
struct CholStep {
  static __device__ __forceinline__ void run(double (&D)[CH_NB], bool& failed) {
    const double piv = readlane_f64(D[K], K);
    const bool ok = piv > 0.0;
    failed |= !ok;
    const double sq = ok ? sqrt(piv) : 1.0;
    const double rinv = 1.0 / sq;
    const double lk = D[K] * rinv;  // lane r > K: L[r][K]; lane K: piv / sq
    const int lane = threadIdx.x;
    D[K] = (lane == K) ? sq : lk;
#pragma unroll
    for (int cc = K + 1; cc < CH_NB; ++cc) D[cc] = fma(-lk, readlane_f64(lk, cc), D[cc]);
    CholStep<K + 1>::run(D, failed);
  }
};
template <>
struct CholStep<CH_NB> {
  static __device__ __forceinline__ void run(double (&)[CH_NB], bool&) {}
};

// Winv_all: [C][64][64], element (k, j) of chain c at c * 4096 + j * 64 + k = (L^-1)[j][k]  (= the B operand of k_panel_rows)
__global__ void __launch_bounds__(64) k_chol_diag(int64_t p, int64_t j0, int nb, double* Qall, double* Winv_all, int* info, long long* bad,
                                                  int64_t chain0) {
  __shared__ double Ls[CH_NB][CH_NB + 1];
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  double* A = Qall + c * p * p + j0 + j0 * p;  // block origin: element (r, cc) at A[r + cc * p]
  double D[CH_NB];
  // beyond the live nb x nb block: the identity (its factor and inverse are the identity: nothing leaks into the live part)
#pragma unroll
  for (int cc = 0; cc < CH_NB; ++cc) D[cc] = (lane < nb && cc < nb) ? ((cc <= lane) ? A[lane + (int64_t)cc * p] : 0.0) : ((cc == lane) ? 1.0 : 0.0);
  bool failed = false;
  CholStep<0>::run(D, failed);
  // the factor goes back (lower triangle of the live block) and into LDS for the inverse
#pragma unroll
  for (int cc = 0; cc < CH_NB; ++cc) {
    const double v = (cc <= lane) ? D[cc] : 0.0;
    Ls[lane][cc] = v;
    if (lane < nb && cc <= lane) A[lane + (int64_t)cc * p] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // column `lane` of W = L^-1 by forward substitution: w_i = (delta_i,lane - sum_{t < i} L[i][t] w_t) / L[i][i]; every lane walks the
  // same (i, t), so the reads of L are wave-wide LDS broadcasts.  (Entries above the diagonal come out as exact zeros.)
  double W[CH_NB];
#pragma unroll
  for (int i = 0; i < CH_NB; ++i) {
    double acc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < i; ++t) acc = fma(-Ls[i][t], W[t], acc);
    W[i] = acc / Ls[i][i];
  }
  double* Wo = Winv_all + c * (int64_t)(CH_NB * CH_NB);
#pragma unroll
  for (int i = 0; i < CH_NB; ++i) Wo[i * CH_NB + lane] = W[i];  // (L^-1)[i][lane] at i * 64 + lane
  if (failed && lane == 0) {
    info[c] = (int)j0 + 1;
    atomicMin((unsigned long long*)bad, (unsigned long long)(chain0 + c));
  }
}

void omc_launch_chol_diag(hipStream_t stream, int64_t Cn, int64_t p, int64_t j0, int nb, double* Qall, double* Winv_all, int* info,
                          long long* bad, int64_t chain0) {
  hipLaunchKernelGGL(k_chol_diag, dim3((unsigned)Cn), dim3(64), 0, stream, p, j0, nb, Qall, Winv_all, info, bad, chain0);
}
