// posdist_kernel.hpp - the two secondary observation modes of obtain_state:
//   a16  Network.get_positional_dist        (network.py:409-430, dist_sign :334-349)
//        State.add_positional_dist: signed TRUE distances to every other vehicle,
//        sorted ascending, divided by the largest distance  -> N-1 values
//   a15  Network.get_positional_dist_piggy  (network.py:432-471)
//        State.add_positional_dist_type == 1: signed table distances (no range
//        filter), inf-norm scaled, np.histogram(v, linspace(-1,1,K+1), weights=v)
//        -> K values.  Explicit edges + weights take NumPy's cumulative path:
//        cw = [0, cumsum(sorted w)] (a SEQUENTIAL float64 prefix sum), bin j =
//        cw[idx_{j+1}] - cw[idx_j] with idx = searchsorted(left; last edge right).
// Both need a per-viewer sort; they run as their own launch right after the fused
// step (same stream; obtain_state follows the step in the reference too) and write
// straight into their sections of the state vector.  One workgroup per env, one
// wave per viewer at a time; the sort is rank-by-counting (ties broken by index,
// which Python's sort of equal floats cannot distinguish anyway), the prefix sum
// is one serial lane - exactness first, these are not the headline path.
#pragma once
#include "common.hpp"
#include "step_kernel.hpp"

namespace diral {

struct PosdistParams {
  int N, A, K, S, NV, NR;
  uint32_t flags;
  int posdist_type, age_limit, out_f64;
  int off_posdist, off_hist;
  const double* pos_x;
  const double* pos_y;
  const uint32_t* tkey;
  const double* tx;
  const double* edges1;     // np.linspace(-1, 1, K+1)
  void* state_out;
};

constexpr int kPdWaves = 4;

__global__ __launch_bounds__(64 * kPdWaves) void posdist_kernel(const PosdistParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = p.N, K = p.K;
  const int NP = (N + 63) & ~63;
  double* s_px = reinterpret_cast<double*>(smem);
  double* s_py = s_px + NP;
  double* s_e1 = s_py + NP;                           // [K+1]
  double* wbase = s_e1 + (K + 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  double* vals = wbase + (size_t)wave * (3 * NP + 2);  // signed values (per wave)
  double* sorted = vals + NP;                          // normalised, in rank order
  double* cw = sorted + NP;                            // [NP+1] prefix sums
  const size_t bN = (size_t)b * N;
  for (int u = tid; u < N; u += blockDim.x) { s_px[u] = p.pos_x[bN + u]; s_py[u] = p.pos_y[bN + u]; }
  for (int j = tid; j <= K; j += blockDim.x) s_e1[j] = p.edges1 ? p.edges1[j] : 0.0;
  __syncthreads();
  const double inf = __builtin_inf();
  const bool full = (p.flags & DIRAL_F_ADD_POSDIST) != 0;
  const bool type1 = (p.flags & DIRAL_F_ADD_POSDIST_PIGGY) && p.posdist_type == 1;

  for (int t = wave; t < N; t += kPdWaves) {
    const double xt = s_px[t], yt = s_py[t];
    const size_t row_out = (bN + t) * (size_t)p.S;
    // ---------------- a16: full-knowledge positional distribution ----------------
    if (full) {
      double dmax = 0.0;
      for (int w = lane; w < NP; w += 64) {
        double v = inf;
        if (w < N && w != t) {
          const double d = dist2d(s_px[w], s_py[w], xt, yt);               // dist(user, tx_user)
          dmax = d > dmax ? d : dmax;
          v = (s_px[w] - xt > 0.0) ? d : -d;                                 // dist_sign(user, tx_user)
        }
        vals[w] = v;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(dmax, off); dmax = o > dmax ? o : dmax; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int w = lane; w < N; w += 64) {
        if (w == t) continue;
        const double v = vals[w];
        int rank = 0;
        for (int q = 0; q < N; ++q) {
          if (q == t) continue;
          const double o = vals[q];
          rank += (o < v || (o == v && q < w)) ? 1 : 0;
        }
        store_out(p.state_out, row_out + p.off_posdist + rank, v / dmax, p.out_f64);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // ---------------- a15: type-1 piggybacked positional histogram ----------------
    if (type1) {
      double dmax = 0.0;
      int nvalid = 0;
      for (int k = lane; k < NP; k += 64) {
        double v = inf;
        if (k < N && k != t) {
          const size_t idx = ((size_t)b * p.NR + k) * p.NV + t;
          const uint32_t w = p.tkey[idx];
          if ((int)(w & 255u) < p.age_limit) {                               // dist_piggy (network.py:538-558)
            const double x1 = p.tx[idx];
            const double y1 = (w >> 8) ? s_py[k] : 0.0;
            const double d = dist2d(x1, y1, xt, yt);
            dmax = d > dmax ? d : dmax;
            v = (x1 - xt > 0.0) ? d : -d;
            nvalid += 1;
          }
        }
        vals[k] = v;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(dmax, off); dmax = o > dmax ? o : dmax;
        nvalid += __shfl_xor(nvalid, off);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int k = lane; k < N; k += 64) {
        const double v = vals[k];
        if (k == t || v == inf) continue;
        int rank = 0;
        for (int q = 0; q < N; ++q) {
          const double o = vals[q];
          if (q == t || o == inf) continue;
          rank += (o < v || (o == v && q < k)) ? 1 : 0;
        }
        sorted[rank] = v / dmax;                                             // dist_sorted / norm
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane == 0) {                                                       // sw.cumsum(), sequential
        double acc = 0.0;
        cw[0] = 0.0;
        for (int i = 0; i < nvalid; ++i) { acc = acc + sorted[i]; cw[i + 1] = acc; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int j = lane; j < K; j += 64) {
        double out = 0.0;
        if (nvalid > 0) {
          const double e_lo = s_e1[j], e_hi = s_e1[j + 1];
          int lo = 0, hi = 0;
          for (int i = 0; i < nvalid; ++i) {
            const double s = sorted[i];
            lo += (s < e_lo) ? 1 : 0;                                        // searchsorted(..., 'left')
            hi += (j + 1 == K) ? ((s <= e_hi) ? 1 : 0) : ((s < e_hi) ? 1 : 0);  // last edge: 'right'
          }
          out = cw[hi] - cw[lo];
        }
        store_out(p.state_out, row_out + p.off_hist + j, out, p.out_f64);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

__host__ inline uint32_t posdist_lds_bytes(int N, int K) {
  const int NP = (N + 63) & ~63;
  return (uint32_t)(8 * (2 * NP + (K + 2) + kPdWaves * (3 * NP + 2)));
}

}  // namespace diral
