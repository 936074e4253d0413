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
// straight into their sections of the state vector.
//   posdist_kernel               any N, any topology: one workgroup per env, one wave per viewer at a
//                                time, rank-by-counting sort (ties broken by index, which Python's sort
//                                of equal floats cannot distinguish anyway), one serial lane for the
//                                prefix sum - the literal reference statement, slow; serves a16 with
//                                vehicles off the common lane only.
//   posdist_sorted_flat_kernel   a16 on the one-lane highway (every pos_y equal - the reference draws
//                                randint(0, 1), network.py:104): the signed distance is a monotone
//                                function of the other vehicle's x, so ONE ranking of the env's x serves
//                                all N viewers.
//   posdist_type1_n64_kernel     a15 for N <= 64: one wave per env, lane = viewer; the viewer's 64
//                                signed distances live in registers, sorted by a fully unrolled bitonic
//                                network of v_min_f64 / v_max_f64; the sequential prefix sum runs in all
//                                64 lanes at once and drops its running value into per-edge LDS slots.
//   posdist_type1_lanes_kernel   a15 for 64 < N <= 256: 4 or 8 lanes per viewer (32 values each), cross-lane bitonic merge.
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
  int do_full, do_type1;    // which of the two modes THIS launch serves
  const double* ring;       // xpos ring (aux_kernels.hpp) when the plane is incomplete, else null
  // the packed table (step_fast64.hpp; step_wide.hpp at N > 128) when it is the current form, else null: the type-1
  // kernels then build every (seq, age) word from the code byte, the age byte and the subject's own sequence number
  // (`tkey` only answers for code-0 entries) instead of waiting for an unpack launch
  const uint32_t* tcode;
  const uint32_t* tage;
  const uint32_t* tseq;
  int flat_y;               // every pos_y == 0: whether a code-0 entry was ever heard (its ypos) cannot matter
};

// viewer u's table word (seq << 8) | age about subject k of env b, from the packed words
__device__ inline uint32_t pd_packed_word(const PosdistParams& p, int b, int k, int u) {
  const size_t qi = ((size_t)b * (p.NR >> 2) + (k >> 2)) * p.NV + u;
  const uint32_t sh = 8u * (uint32_t)(k & 3);
  const uint32_t r = (p.tcode[qi] >> sh) & 255u, a = (p.tage[qi] >> sh) & 255u;
  const size_t row = (size_t)b * p.NR + k;
  // (code 0 = never heard, or older than the codes reach: its xpos comes from the plane either way, and the sequence
  // number only decides the entry's ypos - 0 or the subject's lane - which is 0 on the one-lane highway)
  const uint32_t seq = r ? p.tseq[row] - 8u + (uint32_t)__popc(r) : (p.flat_y ? 0u : (p.tkey[row * p.NV + u] >> 8));
  return (seq << 8) | a;
}

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
  const bool full = p.do_full != 0;
  const bool type1 = p.do_type1 != 0;

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

// dist2d with the general case in line (the kernels below): a call inside an unrolled sweep would force the
// registers of the value array through the calling convention at every call site
__device__ inline double pd_dist(double x1, double y1, double x2, double y2) {
  const double dx = x2 - x1, dy = y2 - y1;
  // |dx| in [2^-500, 2^501) or dx == 0, and dy == 0: sqrt(dx * dx) == |dx| exactly (the same test on the
  // high word as step_fast64.hpp's fast_dist)
  const unsigned int hi = (unsigned int)__double2hiint(dx) & 0x7fffffffu;
  const bool plain = (hi - 0x20b00000u <= 0x3e800000u) || (hi | (unsigned int)__double2loint(dx)) == 0u;
  if (dy == 0.0 && plain) return __hiloint2double((int)hi, __double2loint(dx));
  return __builtin_sqrt(dx * dx + dy * dy);
}

// ---- a16 on a flat highway ------------------------------------------------------------------------
// v(w) = dist_sign(w, t) (network.py:334-349) is +d for x_w > x_t and -d otherwise, d = dist2d a
// monotone function of |x_w - x_t| when the y coordinates agree - so v is monotone in x_w, the sorted
// list of one viewer is the env's x order with the viewer taken out, and the largest distance belongs
// to one of the two extreme vehicles.  Equal x give equal v: their mutual order cannot show.
__global__ __launch_bounds__(256) void posdist_sorted_flat_kernel(const PosdistParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = p.N, NP = (N + 63) & ~63;
  double* s_px = reinterpret_cast<double*>(smem);
  double* s_py = s_px + NP;
  double* s_dmax = s_py + NP;
  int* s_rank = reinterpret_cast<int*>(s_dmax + NP);
  int* s_ord = s_rank + NP;
  const int tid = threadIdx.x, b = blockIdx.x;
  const size_t bN = (size_t)b * N;
  for (int u = tid; u < N; u += 256) { s_px[u] = p.pos_x[bN + u]; s_py[u] = p.pos_y[bN + u]; }
  __syncthreads();
  for (int w = tid; w < N; w += 256) {
    const double x = s_px[w];
    int r = 0;
    for (int q = 0; q < N; ++q) { const double o = s_px[q]; r += (o < x || (o == x && q < w)) ? 1 : 0; }
    s_rank[w] = r; s_ord[r] = w;
  }
  __syncthreads();
  if (N < 2) return;
  const int lo = s_ord[0], hi = s_ord[N - 1], M = N - 1;
  for (int t = tid; t < N; t += 256) {                                   // the farthest vehicle is one of the two extremes
    const double dlo = dist2d(s_px[lo], s_py[lo], s_px[t], s_py[t]), dhi = dist2d(s_px[hi], s_py[hi], s_px[t], s_py[t]);
    s_dmax[t] = dlo > dhi ? dlo : dhi;
  }
  __syncthreads();
  // one wave per viewer at a time, lanes along its sorted list: the viewer's own values stay in registers,
  // the row leaves coalesced
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < N; t += 4) {
    const double xt = s_px[t], yt = s_py[t], dmax = s_dmax[t];
    const int rk = s_rank[t];
    const size_t row = (bN + t) * (size_t)p.S + p.off_posdist;
    for (int r = lane; r < M; r += 64) {
      const int w = s_ord[r + (r >= rk ? 1 : 0)];                          // the viewer itself is skipped
      const double xw = s_px[w];
      const double d = pd_dist(xw, s_py[w], xt, yt);
      const double v = (xw - xt > 0.0) ? d : -d;
      store_out(p.state_out, row + r, v / dmax, p.out_f64);
    }
  }
}

__host__ inline uint32_t posdist_flat_lds_bytes(int N) {
  const int NP = (N + 63) & ~63;
  return (uint32_t)(NP * (8 + 8 + 8 + 4 + 4));
}

// ---- a15, N <= 64 -----------------------------------------------------------------------------------
__device__ inline double pd_readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

constexpr int kPd1Stride = 65;                         // doubles per row of edge sums: lane t at column t, rows 2 banks apart

__host__ __device__ inline uint32_t posdist_type1_lds_bytes(int K) { return (uint32_t)(8 * (66 + (K + 2) * kPd1Stride)); }

__global__ __launch_bounds__(64, 3) void posdist_type1_n64_kernel(const PosdistParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* const s_e1 = reinterpret_cast<double*>(smem);                  // [K + 1] edges
  double* const s_c = s_e1 + 66;                                         // [K + 2][kPd1Stride] edge sums C_j per viewer (+ a slot for the fillers)
  const int N = p.N, K = p.K, lane = threadIdx.x, b = blockIdx.x;
  const size_t bN = (size_t)b * N;
  const bool live = lane < N;
  const double xt = live ? p.pos_x[bN + lane] : 0.0, yt = live ? p.pos_y[bN + lane] : 0.0;
  for (int j = lane; j <= K; j += 64) s_e1[j] = p.edges1[j];
  const double inf = __builtin_inf();

  // the viewer's signed table distances (dist_piggy, network.py:538-558): row k of the subject-major
  // table holds what every viewer knows about k - one coalesced row read per subject, in batches of
  // three straight unrolled sweeps (table words, xpos, arithmetic) that keep 16 + 16 loads in flight.
  // (rows past N - 1 are read all the same - the table allocations carry 64 rows of slack, diral_env_create -
  // and masked out: every row sits at a compile-time offset from one base pointer)
  double v[64];
  double dmax = 0.0;
  int nvalid = 0;
  const uint32_t ts_lane = p.tseq ? p.tseq[(size_t)b * p.NR + (lane < p.NR ? lane : 0)] : 0u;
  const uint32_t* const trow = p.tkey + (size_t)b * p.NR * 64 + lane;     // NV == 64 for N <= 64
  const double* const xrow = p.tx + (size_t)b * p.NR * 64 + lane;
  const double* const rrow = p.ring ? p.ring + (size_t)b * p.NR * 8 : nullptr;
#pragma unroll
  for (int k0 = 0; k0 < 64; k0 += 16) {                                    // 16 rows per batch: 48 registers in flight
    uint32_t tw[16];
    // (the packed table on the one-lane highway: every word is rebuilt from the code / age words below - a code-0 entry's
    // sequence number only decides its ypos -, the plane is not read at all)
    if (!(p.tcode && p.flat_y)) {
#pragma unroll
      for (int c = 0; c < 16; ++c) tw[c] = trow[(k0 + c) * 64];
    }
    if (p.tcode) {
      // the packed table: four code words + four age words serve the batch's 16 rows; the subjects' own sequence
      // numbers sit in `ts_lane` (lane k: subject k)
      uint32_t cq[4], aq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int qi = (k0 >> 2) + q;
        const size_t at = ((size_t)b * (p.NR >> 2) + (qi < (p.NR >> 2) ? qi : 0)) * 64 + lane;
        cq[q] = p.tcode[at];
        aq[q] = p.tage[at];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t sh = 8u * (uint32_t)(c & 3);
        const uint32_t r = (cq[c >> 2] >> sh) & 255u, a = (aq[c >> 2] >> sh) & 255u;
        const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)ts_lane, k0 + c);
        // (code 0: never heard, or older than the codes reach - the plane holds its xpos either way, and its sequence
        // number only decides the ypos, 0 on the one-lane highway)
        const uint32_t seq = r ? tk - 8u + (uint32_t)__popc(r) : (p.flat_y ? 0u : (tw[c] >> 8));
        tw[c] = (seq << 8) | a;
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = k0 + c;
      const uint32_t seq = tw[c] >> 8;
      const double* src = xrow + k * 64;
      if (rrow && k < N) {                                                 // uniform: the plane holds only entries 7+ stamps old
        const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)tw[c], k) >> 8;   // row k's diagonal
        if (tk - seq <= 7u) src = rrow + k * 8 + (seq & 7u);
      }
      v[k] = *src;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = k0 + c;
      const uint32_t w = tw[c];
      const double x1 = v[k];
      const bool valid = live && k < N && k != lane && (int)(w & 255u) < p.age_limit;
      const double yk = pd_readlane_f64(yt, k);
      const double y1 = (w >> 8) ? yk : 0.0;
      const double d = pd_dist(x1, y1, xt, yt);
      dmax = (valid && d > dmax) ? d : dmax;
      v[k] = valid ? ((x1 - xt > 0.0) ? d : -d) : inf;
      nvalid += valid ? 1 : 0;
      asm volatile("" : "+v"(v[k]), "+v"(dmax), "+v"(nvalid));            // finish the row here (its table word and xpos die)
    }
    // keep the batches apart (registers): no load of the next batch may start above this point
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  // dist_sorted / norm before the sort instead of after it: dividing by the positive norm keeps the order
  // and yields the same 64 quotients (inf stays inf; all-zero distances give the reference's NaN)
#pragma unroll
  for (int k = 0; k < 64; ++k) v[k] = v[k] / dmax;
  // ascending bitonic network, compile-time indices: everything stays in registers.  Invalid entries are
  // +inf and end up behind the nvalid real ones; equal values need no tie rule.
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const double a = v[i], c = v[l];
          const double mn = __builtin_fmin(a, c), mx = __builtin_fmax(a, c);
          const bool asc = (i & k2) == 0;
          v[i] = asc ? mn : mx;
          v[l] = asc ? mx : mn;
        }
      }
    }
  }
  // np.histogram(v, linspace(-1, 1, K + 1), weights = v): cw = [0, cumsum(sorted)] - a sequential sum - and
  // bin j = cw[#(s < e_{j+1})] - cw[#(s < e_j)], the last edge counted with <=.  C_j = cw[#(s < e_j)] is the
  // running sum through the last value below edge j.  No search and no loop over edges: every value writes the
  // running sum (itself included) into the slot of the first edge above it - c = #(edges <= s), from the
  // uniform spacing and two compares against the real edges - later values overwrite earlier ones, and a
  // forward fill over the K + 1 slots completes the edges no value sits directly below.  (Values never
  // exceed e_K = 1 = |v| / max |v|, so C_K is the total; slot 0 stays empty: nothing is below -1.)
  __syncthreads();                                                       // s_e1
  double* const col = s_c + lane;
  const int kUnsetHi = 0x7ff8dead;                                       // a NaN no sum can be
  for (int j = 0; j <= K + 1; ++j) col[j * kPd1Stride] = __hiloint2double(kUnsetHi, 0);   // slots 0..K, + one for the fillers
  const int nreal = dmax > 0.0 ? nvalid : 0;                             // all distances 0: the reference's NaNs fall into no bin
  const double half_k = 0.5 * (double)K;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const bool real = i < nreal;                                         // (the +inf fillers behind the real values take no part)
    const double s = real ? v[i] : 0.0;
    int est = (int)((s + 1.0) * half_k);
    est = est < 0 ? 0 : (est > K - 1 ? K - 1 : est);
    const double e0 = s_e1[est], e1 = s_e1[est + 1];
    const int c = est + 1 - (s < e0 ? 1 : 0) + ((est + 1 < K && !(s < e1)) ? 1 : 0);
    acc = real ? acc + s : acc;
    col[(real ? c : K + 1) * kPd1Stride] = acc;
  }
  double cur = 0.0;
  for (int j = 0; j <= K; ++j) {
    const double t = col[j * kPd1Stride];
    cur = __double2hiint(t) == kUnsetHi ? cur : t;
    col[j * kPd1Stride] = cur;
  }
  __syncthreads();
  // rows leave coalesced: consecutive lanes on consecutive bins of a viewer
  for (int e = lane; e < N * K; e += 64) {
    const int t = e / K, j = e - t * K;
    const double out = s_c[(j + 1) * kPd1Stride + t] - s_c[j * kPd1Stride + t];
    store_out(p.state_out, (bN + t) * (size_t)p.S + p.off_hist + j, out, p.out_f64);
  }
}

// ---- a15, 64 < N <= 256: LPV lanes per viewer, VL subjects per lane -----------------------------------
// The same kernel with a viewer's table spread over LPV neighbouring lanes, VL subjects each (lane l: viewer
// l / LPV of the wave's 64 / LPV, subjects VL (l % LPV) ...).  Every lane sorts its VL values as above; the
// lanes of a viewer then merge their runs with the cross-lane steps of the same bitonic network - partner
// values through DPP moves (quad permutes, row_half_mirror), the low lane keeps the minima - after which lane s
// holds ranks VL s ... VL s + VL - 1 of the viewer's sorted list.  The sequential prefix sum passes from lane to lane (LPV
// passes over the same code, one lane of each viewer active per pass).
template <int CTRL>
__device__ inline double pd_quad_perm(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kQuadXor3 = 0x1B, kQuadShr1 = 0x90;   // quad_perm [1,0,3,2] [2,3,0,1] [3,2,1,0] [0,0,1,2]
constexpr int kRowHalfMirror = 0x141, kRowShr1 = 0x111;                                 // lane s <-> 7 - s within a row half; lane s <- s - 1 within a row of 16
#ifndef DIRAL_TYPE1_CUMSUM_INPLACE
#define DIRAL_TYPE1_CUMSUM_INPLACE 1     // posdist_type1_lanes_kernel: the sequential sum by the lane whose turn it is only, one store per slot boundary
#endif
#ifndef DIRAL_TYPE1_MINWAVES
#define DIRAL_TYPE1_MINWAVES 4           // posdist_type1_lanes_kernel<LPV, 32>: waves per SIMD the register allocation is held to
#endif

__device__ inline double pd_pick(bool upper, double a, double b) {        // the partner keeps the other one
  const double mn = __builtin_fmin(a, b), mx = __builtin_fmax(a, b);
  return upper ? mx : mn;
}

template <int LPV, int VL>
__global__ __launch_bounds__(64, VL == 64 ? 2 : DIRAL_TYPE1_MINWAVES) void posdist_type1_lanes_kernel(const PosdistParams p) {
  // VL subjects per lane, LPV lanes per viewer: <2, 64> / <4, 64> (rounds 3-4: 128 VGPRs of values, two waves per SIMD - the
  // kernel waited for its own round trips) or <4, 32> / <8, 32> (N <= 128 / 256: half the values per lane, one more level of
  // cross-lane merging, twice the waves)
  static_assert((LPV == 2 || LPV == 4 || LPV == 8) && (VL == 32 || VL == 64), "a viewer's lanes share a row of 8");
  constexpr int VW = 64 / LPV;                                           // viewers per wave
  constexpr int ST = VW | 1;                                             // doubles per row of edge sums (one column per viewer of the wave)
  extern __shared__ __align__(16) unsigned char smem[];
  double* const s_e1 = reinterpret_cast<double*>(smem);                  // [K + 1] edges
  double* const s_c = s_e1 + 66;                                         // [K + 2][ST] edge sums per viewer (columns 0 .. VW - 1)
  double* const s_py = s_c + (p.K + 2) * ST;                     // [VL LPV] pos_y of the env
  uint32_t* const s_ts = reinterpret_cast<uint32_t*>(s_py + VL * LPV);   // [VL LPV] the subjects' own sequence numbers
  const int N = p.N, K = p.K, NV = p.NV, lane = threadIdx.x;
  const int nvb = (N + VW - 1) / VW;                                     // viewer blocks per env
  const int b = blockIdx.x / nvb, vb = blockIdx.x - b * nvb;
  const int sub = lane & (LPV - 1), vw = lane / LPV;
  const int t = vb * VW + vw;
  const size_t bN = (size_t)b * N;
  const bool live = t < N;
  const double xt = live ? p.pos_x[bN + t] : 0.0, yt = live ? p.pos_y[bN + t] : 0.0;
  for (int j = lane; j <= K; j += 64) s_e1[j] = p.edges1[j];
  // (the one-lane highway: every y is 0; the subjects' own numbers - `tseq`, or the diagonal of the plane - once per wave
  // into LDS: two global loads per ENTRY before)
  for (int u = lane; u < VL * LPV; u += 64) {
    s_py[u] = (u < N && !p.flat_y) ? p.pos_y[bN + u] : 0.0;
    const size_t ru = (size_t)b * p.NR + (u < p.NR ? u : 0);
    s_ts[u] = p.tseq ? p.tseq[ru] : p.tkey[ru * NV + (u < p.NR ? u : 0)] >> 8;
  }
  __syncthreads();
  const double inf = __builtin_inf();

  double v[VL];
  double dmax = 0.0;
  int nvalid = 0;
  const size_t row0 = (size_t)b * p.NR + sub * VL;                        // the lane's first subject row
  const uint32_t* const trow = p.tkey + row0 * NV + t;
  const double* const xrow = p.tx + row0 * NV + t;
  const double* const rrow = p.ring ? p.ring + row0 * 8 : nullptr;
#pragma unroll
  for (int k0 = 0; k0 < VL; k0 += 16) {
    uint32_t tw[16];
    if (!(p.tcode && p.flat_y)) {                                          // (as in posdist_type1_n64_kernel)
#pragma unroll
      for (int c = 0; c < 16; ++c) tw[c] = trow[(size_t)(k0 + c) * NV];
    }
    if (p.tcode) {
      // the packed table: four code words + four age words serve the batch's 16 rows (see posdist_type1_n64_kernel)
      uint32_t cq[4], aq[4];
      const int tq = t < NV ? t : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int qi = ((sub * VL + k0) >> 2) + q;
        const size_t at = ((size_t)b * (p.NR >> 2) + (qi < (p.NR >> 2) ? qi : 0)) * NV + tq;
        cq[q] = p.tcode[at];
        aq[q] = p.tage[at];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int kg = sub * VL + k0 + c;
        const uint32_t sh = 8u * (uint32_t)(c & 3);
        const uint32_t r = (cq[c >> 2] >> sh) & 255u, a = (aq[c >> 2] >> sh) & 255u;
        const uint32_t tk = s_ts[kg];
        const uint32_t seq = r ? tk - 8u + (uint32_t)__popc(r) : (p.flat_y ? 0u : (tw[c] >> 8));
        tw[c] = (seq << 8) | a;
      }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = k0 + c;
      const uint32_t seq = tw[c] >> 8;
      const double* src = xrow + (size_t)k * NV;
      if (rrow) {                                                          // uniform: the plane holds only entries 7+ stamps old
        const uint32_t tk = s_ts[sub * VL + k];
        if (sub * VL + k < N && tk - seq <= 7u) src = rrow + k * 8 + (seq & 7u);
      }
      v[k] = *src;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int k = k0 + c, kg = sub * VL + k;
      const uint32_t w = tw[c];
      const double x1 = v[k];
      const bool valid = live && kg < N && kg != t && (int)(w & 255u) < p.age_limit;
      const double y1 = (w >> 8) ? s_py[kg] : 0.0;
      const double d = pd_dist(x1, y1, xt, yt);
      dmax = (valid && d > dmax) ? d : dmax;
      v[k] = valid ? ((x1 - xt > 0.0) ? d : -d) : inf;
      nvalid += valid ? 1 : 0;
      asm volatile("" : "+v"(v[k]), "+v"(dmax), "+v"(nvalid));
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  // norm and count of the whole viewer
  {
    double o = pd_quad_perm<kQuadXor1>(dmax);
    dmax = o > dmax ? o : dmax;
    nvalid += __builtin_amdgcn_mov_dpp(nvalid, kQuadXor1, 0xf, 0xf, true);
    if constexpr (LPV >= 4) {
      o = pd_quad_perm<kQuadXor2>(dmax);
      dmax = o > dmax ? o : dmax;
      nvalid += __builtin_amdgcn_mov_dpp(nvalid, kQuadXor2, 0xf, 0xf, true);
    }
    if constexpr (LPV == 8) {                                              // the other quad of the viewer's eight lanes
      o = pd_quad_perm<kRowHalfMirror>(dmax);
      dmax = o > dmax ? o : dmax;
      nvalid += __builtin_amdgcn_mov_dpp(nvalid, kRowHalfMirror, 0xf, 0xf, true);
    }
  }
#pragma unroll
  for (int k = 0; k < VL; ++k) v[k] = v[k] / dmax;
  // every lane: its VL values ascending
#pragma unroll
  for (int k2 = 2; k2 <= VL; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < VL; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const double a = v[i], c = v[l];
          const double mn = __builtin_fmin(a, c), mx = __builtin_fmax(a, c);
          const bool asc = (i & k2) == 0;
          v[i] = asc ? mn : mx;
          v[l] = asc ? mx : mn;
        }
      }
    }
  }
  // a lane's 64 values are a bitonic sequence: ascending by the last six steps of the network
  auto merge_in_lane = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = VL / 2; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < VL; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const double a = v[i], c = v[l];
          v[i] = __builtin_fmin(a, c);
          v[l] = __builtin_fmax(a, c);
        }
      }
    }
  };
  // two ascending runs X (low lanes) and Y (high lanes): min(X[i], Y[n-1-i]) stays with X, the max goes to Y's
  // place n-1-i - both halves bitonic, every low value <= every high one
  const bool odd = (sub & 1) != 0;
#pragma unroll
  for (int i = 0; i < VL / 2; ++i) {
    const double t1 = pd_quad_perm<kQuadXor1>(v[VL - 1 - i]), t2 = pd_quad_perm<kQuadXor1>(v[i]);
    v[i] = pd_pick(odd, v[i], t1);
    v[VL - 1 - i] = pd_pick(odd, v[VL - 1 - i], t2);
    asm volatile("" : "+v"(v[i]), "+v"(v[VL - 1 - i]));                   // (pair by pair: the permuted copies must not pile up)
  }
  merge_in_lane();
  const bool hi2 = (sub & 2) != 0;
  if constexpr (LPV >= 4) {
    // runs of 2 VL: lanes 0,1 against lanes 3,2
#pragma unroll
    for (int i = 0; i < VL / 2; ++i) {
      const double t1 = pd_quad_perm<kQuadXor3>(v[VL - 1 - i]), t2 = pd_quad_perm<kQuadXor3>(v[i]);
      v[i] = pd_pick(hi2, v[i], t1);
      v[VL - 1 - i] = pd_pick(hi2, v[VL - 1 - i], t2);
      asm volatile("" : "+v"(v[i]), "+v"(v[VL - 1 - i]));
    }
#pragma unroll
    for (int i = 0; i < VL; ++i) {                                         // stride VL of the 2 VL-sequence
      v[i] = pd_pick(odd, v[i], pd_quad_perm<kQuadXor1>(v[i]));
      asm volatile("" : "+v"(v[i]));
    }
    merge_in_lane();
  }
  if constexpr (LPV == 8) {
    // runs of 4 VL: lanes 0..3 against lanes 7..4 (row_half_mirror: lane s <-> 7 - s), then the strides 2 VL and VL
    const bool hi3 = (sub & 4) != 0;
#pragma unroll
    for (int i = 0; i < VL / 2; ++i) {
      const double t1 = pd_quad_perm<kRowHalfMirror>(v[VL - 1 - i]), t2 = pd_quad_perm<kRowHalfMirror>(v[i]);
      v[i] = pd_pick(hi3, v[i], t1);
      v[VL - 1 - i] = pd_pick(hi3, v[VL - 1 - i], t2);
      asm volatile("" : "+v"(v[i]), "+v"(v[VL - 1 - i]));
    }
#pragma unroll
    for (int i = 0; i < VL; ++i) {
      v[i] = pd_pick(hi2, v[i], pd_quad_perm<kQuadXor2>(v[i]));
      asm volatile("" : "+v"(v[i]));
    }
#pragma unroll
    for (int i = 0; i < VL; ++i) {
      v[i] = pd_pick(odd, v[i], pd_quad_perm<kQuadXor1>(v[i]));
      asm volatile("" : "+v"(v[i]));
    }
    merge_in_lane();
  }
  // lane `sub` now holds ranks VL sub .. VL sub + VL - 1 of the viewer's list; the histogram as in the one-lane kernel
  double* const col = s_c + vw;
  const int kUnsetHi = 0x7ff8dead;
  if (sub == 0)
    for (int j = 0; j <= K + 1; ++j) col[j * ST] = __hiloint2double(kUnsetHi, 0);
  const int nreal_all = dmax > 0.0 ? nvalid : 0;
  const int nreal = nreal_all - VL * sub;                                  // of this lane's VL ranks (<= 0: none)
  const double half_k = 0.5 * (double)K;
  // the slot of every value first, all lanes at once (four 8-bit slot numbers per register; the fillers get
  // the spare slot and the value 0) - the passes below only add and store
  uint32_t cpk[VL / 4];
#pragma unroll
  for (int q = 0; q < VL / 4; ++q) cpk[q] = 0u;
#pragma unroll
  for (int i = 0; i < VL; ++i) {
    const bool real = i < nreal;
    const double s = real ? v[i] : 0.0;
    v[i] = s;
    int est = (int)((s + 1.0) * half_k);
    est = est < 0 ? 0 : (est > K - 1 ? K - 1 : est);
    const double e0 = s_e1[est], e1 = s_e1[est + 1];
    const int c = est + 1 - (s < e0 ? 1 : 0) + ((est + 1 < K && !(s < e1)) ? 1 : 0);
    cpk[i >> 2] |= (uint32_t)(real ? c : K + 1) << (8 * (i & 3));
  }
#if DIRAL_TYPE1_CUMSUM_INPLACE
  // NumPy's cumsum in its own order (cw[j + 1] = cw[j] + sa[j], numpy/lib/_histograms_impl.py): a viewer's LPV lanes take
  // turns, lowest ranks first, each continuing from the total the lane below ended with.  The histogram needs the running
  // sum behind the LAST value of every slot only (searchsorted + diff): that value - and no other, and only in its lane's
  // own pass - stores its sum; the wave used to store every value's sum in every pass (the other passes' into a spare
  // slot: four instructions and an LDS store per value and pass).  Last of its slot = the slot number changes behind it;
  // behind a lane's last value comes the first of the lane above (the viewer's top lane: the end of the list).
  // (the sums are not kept: in-place updates of the values - behind a lane test or as selects - cost 172 spilled registers)
  unsigned int chg[VL / 4];                                                // byte i != 0: value i is the last of its slot
  {
    const unsigned int nxt0 = (unsigned int)__builtin_amdgcn_mov_dpp((int)cpk[0], 0x101 /* row_shl:1 */, 0xf, 0xf, true);
    const unsigned int next_first = sub == LPV - 1 ? 0xffu : (nxt0 & 255u);
#pragma unroll
    for (int q = 0; q < VL / 4; ++q) {
      const unsigned int nb = q + 1 < VL / 4 ? cpk[q + 1 < VL / 4 ? q + 1 : q] : next_first;
      chg[q] = cpk[q] ^ ((cpk[q] >> 8) | (nb << 24));
    }
  }
  double acc = 0.0;
#pragma unroll 1
  for (int ph = 0; ph < LPV; ++ph) {
    const double before = pd_quad_perm<kRowShr1>(acc);                     // the running sum of the lane below, complete by now
    const bool active = sub == ph;
    if (active) acc = ph > 0 ? before : 0.0;
#pragma unroll
    for (int i = 0; i < VL; ++i) {
      acc = acc + v[i];
      if (active && ((chg[i >> 2] >> (8 * (i & 3))) & 255u) != 0u) {
        const int c = (int)((cpk[i >> 2] >> (8 * (i & 3))) & 255u);
        col[c * ST] = acc;
      }
    }
  }
#else
  double acc = 0.0;
#pragma unroll 1
  for (int ph = 0; ph < LPV; ++ph) {
    const double before = pd_quad_perm<kRowShr1>(acc);                     // the running sum of the lane below, complete by now
    const bool active = sub == ph;
    if (active) acc = ph > 0 ? before : 0.0;                               // (what a lane adds outside its own pass goes to the spare slot)
#pragma unroll
    for (int i = 0; i < VL; ++i) {
      acc = acc + v[i];
      const int c = (int)((cpk[i >> 2] >> (8 * (i & 3))) & 255u);
      col[(active ? c : K + 1) * ST] = acc;
    }
  }
#endif
  if (sub == 0) {
    double cur = 0.0;
    for (int j = 0; j <= K; ++j) {
      const double tt = col[j * ST];
      cur = __double2hiint(tt) == kUnsetHi ? cur : tt;
      col[j * ST] = cur;
    }
  }
  __syncthreads();
  for (int e = lane; e < VW * K; e += 64) {
    const int tv = e / K, j = e - tv * K, to = vb * VW + tv;
    if (to < N) {
      const double out = s_c[(j + 1) * ST + tv] - s_c[j * ST + tv];
      store_out(p.state_out, (bN + to) * (size_t)p.S + p.off_hist + j, out, p.out_f64);
    }
  }
}

__host__ __device__ inline uint32_t posdist_type1_lanes_lds_bytes(int K, int npad, int lpv) {
  return 8u * (66u + (uint32_t)(K + 2) * (uint32_t)((64 / lpv) | 1) + (uint32_t)npad) + 4u * (uint32_t)npad;
}

__host__ inline uint32_t posdist_lds_bytes(int N, int K) {
  const int NP = (N + 63) & ~63;
  return (uint32_t)(8 * (2 * NP + (K + 2) + kPdWaves * (3 * NP + 2)));
}

}  // namespace diral
