// step_kernel.hpp - the fused env-step kernel (one workgroup = one env).
//
// Replaces, for one env per workgroup and one time-slot per launch:
//   TestEnv.my_step / my_step_ch / my_step_design  (test_env.py:124-266/351-443/269-349)
//   + TestEnv.obtain_state                          (test_env.py:527-583)
// i.e. Network.periodic_update, find_closest_tx, received_update,
// calculate_reward_weights, update_positions, get_positional_dist_2_piggy.
//
// Mapping to CDNA4 (no MFMA: there is no dense contraction on this path):
//   lane  = viewer vehicle u (VPL viewers per lane when N > 64)
//   wave  = 16 subject columns of the neighbour table
//   * per-resource transmitter sets are 64-bit wave ballots; collision counts
//     and the PRR in-range/received counts are popcounts of ballots
//   * the gossip merge Vehicle.received_update is a per-COLUMN problem: for a
//     fixed subject k the entry is fully determined by its sequence number, so
//     "rx copies tx's row where tx is fresher" == key[u] = max(key[u], key[m_i(u)])
//     with key = (seq << 8) | source-viewer, applied for resources i = 0..A-1 in
//     order.  Columns are independent, so a wave walks all A
//     resources for its 16 columns with NO workgroup barrier:
//       N <= 64 : the gather is one ds_bpermute per column per resource,
//       N  > 64 : the column lives in a wave-private LDS scratch (in-order LDS
//                 queue of a wave orders its own writes and reads).
//     xpos follows afterwards with ONE gather from the recorded source viewer.
//   * HBM traffic per env-slot: each table word read once, written once,
//     coalesced along the viewer axis (subject-major layout, common.hpp); for
//     N <= 64 the whole table read is issued before any compute so its latency
//     hides behind the closest-transmitter phase.
//
// float64 everywhere the reference uses Python floats; build with
// -ffp-contract=off so a*b+c is never fused (bin edges, distances).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace diral {

#ifdef DIRAL_TIMING
#define DIRAL_STAMP(i) do { if (lane == 0 && p.dbg) p.dbg[((size_t)b * WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DIRAL_STAMP(i) do {} while (0)
#endif
#ifndef DIRAL_MINWAVES
#define DIRAL_MINWAVES 1
#endif
#ifndef DIRAL_PREFETCH
#define DIRAL_PREFETCH 1
#endif
#ifndef DIRAL_PACKED_WIDE
#define DIRAL_PACKED_WIDE 1            // N > 64: 16-bit packed gossip merge (exact; falls back per pass)
#endif

template <int VPL>
struct Geo {
  static constexpr int NPAD = 64 * VPL;      // padded viewer count
  static constexpr int WAVES = 4 * VPL;      // 16 subject columns per wave
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int CC = 16 / VPL;        // columns per register chunk (16 key regs)
  static constexpr int NCH = VPL;            // chunks per wave
};

// Network.dist (network.py:318-332): sqrt(dx^2 + dy^2) with correctly rounded
// squares (DESIGN.md on the reference's pow()).  When dy == 0 - every pair in a
// random topology, where all y are 0 - sqrt(fl(dx*dx)) == |dx| exactly in IEEE
// binary64 (no over/underflow for 2^-500 <= |dx| <= 2^500), so the sqrt is skipped.
// rare general paths are kept out of line so the hot code stays compact
// (the fused kernels were instruction-fetch bound when these were inlined)
__device__ DIRAL_OUTLINE double dist_general(double dx, double dy) {
  return __builtin_sqrt(dx * dx + dy * dy);
}
__device__ inline double dist2d(double x1, double y1, double x2, double y2) {
  const double dx = x2 - x1, dy = y2 - y1;
  const double ax = __builtin_fabs(dx);
  // (dx == 0 as well: sqrt(0) == 0 - a vehicle measured against itself, as the transmitter search does)
  if (dy == 0.0 && (ax == 0.0 || (ax >= 0x1p-500 && ax <= 0x1p500))) return ax;
  return dist_general(dx, dy);
}

// dist2d for the out-of-line reward functions: the IEEE sqrt is inlined so that they stay LEAF functions - a nested
// call makes the callee save its return address through a callee-saved VGPR in scratch memory, and a kernel that
// may reach such a callee carries a private segment (16 B / lane) for every wave it launches
__device__ inline double dist2d_leaf(double x1, double y1, double x2, double y2) {
  const double dx = x2 - x1, dy = y2 - y1;
  const double ax = __builtin_fabs(dx);
  if (dy == 0.0 && (ax == 0.0 || (ax >= 0x1p-500 && ax <= 0x1p500))) return ax;
  return __builtin_sqrt(dx * dx + dy * dy);
}

// Bin of a value of the type-2 histogram, np.histogram(v, K, range=(-Rb, Rb)) (network.py:500): NumPy estimates the bin from
// (v - first) / (last - first) * K and corrects it by at most one step against the float edges.  The correction can
// only act when v lies within rounding distance of an edge.  With t = (v + Rb) * inv_w: t, the edges (linspace) and the
// comparisons are each within 2^-45 bin widths of their exact values (K <= 64), so if the fractional part of t lies in
// [2^-20, 1 - 2^-20] the value is safely inside bin floor(t) and the edges need not be read - one LDS round trip and two
// f64 compares less per table entry.  `unsafe` lanes (an exact hit of an edge: integer-valued positions) take the reads.
// Requires -Rb <= v < Rb (so 0 <= t <= K; t == K rounds in from below: fractional part 0, unsafe - the edge branch
// clamps the estimate to K - 1 first).
__device__ inline int hist_bin_estimate(double v, double Rb, double inv_w, int K, bool& unsafe) {
  const double t = (v + Rb) * inv_w;
  const int est = (int)t;                        // (t == K: fractional part 0 - the caller's edge branch clamps, hist_bin_clamp)
  const double fr = __builtin_amdgcn_fract(t);
  unsafe = !(__builtin_fabs(fr - 0.5) <= 0.5 - 0x1p-20);
  return est;
}
__device__ inline int hist_bin_clamp(int est, int K) { return est > K - 1 ? K - 1 : est; }

// Python float `%` for the position wrap (network.py:203): fast exact path when
// 0 <= s <= 2L (Sterbenz), generic fmod + sign fix-up otherwise.
__device__ DIRAL_OUTLINE double py_mod_general(double s, double L) {
  double m = fmod(s, L);
  if (m != 0.0) { if ((L < 0) != (m < 0)) m += L; } else { m = copysign(0.0, L); }
  return m;
}
__device__ inline double py_mod_pos(double s, double L) {
  if (s >= 0.0 && s < L) return s;
  if (s >= L && s <= 2.0 * L) {
    const double r = s - L;            // exact
    return (r >= L) ? r - L : r;       // s == 2L -> 0
  }
  return py_mod_general(s, L);
}

// np.histogram uniform-bin index (numpy/lib/_histograms_impl.py fast path).
// NumPy estimates the index with a division and then corrects it against the
// actual edges ("not guaranteed to give exactly consistent results within ~1
// ULP of the bin edges"), so its result is THE bin with edges[i] <= v <
// edges[i+1].  Any estimate followed by the same edge correction lands in the
// same bin; a reciprocal multiply replaces the f64 division.  `edges` are the
// exact np.linspace values (strictly increasing, checked at create).
__device__ inline int hist_bin(double v, double first, double inv_width, int K, const double* edges) {
  int idx = (int)((v - first) * inv_width);
  idx = idx < 0 ? 0 : (idx > K - 1 ? K - 1 : idx);
  while (idx > 0 && v < edges[idx]) --idx;
  while (idx < K - 1 && v >= edges[idx + 1]) ++idx;
  return idx;
}

__device__ inline void store_out(void* base, size_t idx, double v, int f64) {
  if (f64) reinterpret_cast<double*>(base)[idx] = v;
  else reinterpret_cast<float*>(base)[idx] = (float)v;
}

// Orders this wave's LDS accesses for the COMPILER only.  The hardware already
// executes one wave's DS instructions in issue order, so a later ds_read sees an
// earlier ds_write of the same wave without any wait; a real fence would drain
// lgkmcnt at every merge step (measured: the dominant cost at N > 64).
__device__ inline void wave_lds_order() { asm volatile("" ::: "memory"); }

// wave-uniform 64-bit value -> SGPR pair, so branches on it are scalar
__device__ inline unsigned long long uniform_u64(unsigned long long v) {
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
  const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ inline int popc_masks(const unsigned long long* m, int vpl) {
  int c = 0;
  for (int j = 0; j < vpl; ++j) c += __popcll(m[j]);
  return c;
}

// Network.calculate_reward_weights (network.py:273-300) for the tx set `mk`,
// evaluated wave-uniformly (every lane computes the same value).
template <int VPL>
__device__ inline int reward_weight(const StepParams& p, const unsigned long long (&mk)[VPL],
                                    const double* s_px, const double* s_py) {
  // calculate_avg_distance (network.py:307-316): combinations order, serial sum
  double s = 0.0;
  int cnt = 0;
  for (int ja = 0; ja < VPL; ++ja) {
    unsigned long long ma = mk[ja];
    while (ma) {
      const int a = ja * 64 + __builtin_ctzll(ma);
      ma &= ma - 1;
      for (int jb = ja; jb < VPL; ++jb) {
        unsigned long long mb = (jb == ja) ? ma : mk[jb];
        while (mb) {
          const int b = jb * 64 + __builtin_ctzll(mb);
          mb &= mb - 1;
          s = s + dist2d(s_px[a], s_py[a], s_px[b], s_py[b]);
          ++cnt;
        }
      }
    }
  }
  const double m = s / (double)cnt;
  if (p.flags & DIRAL_F_TOY_WEIGHTS) {
    // calculate_norm (network.py:225-246)
    double x_min = p.L + 1, x_max = -p.L - 1;
    int umin = 0, umax = 0;
    for (int u = 0; u < p.N; ++u) {
      const double x = s_px[u];
      if (x < x_min) { x_min = x; umin = u; }
      if (x > x_max) { x_max = x; umax = u; }
    }
    const double norm = dist2d(s_px[umin], s_py[umin], s_px[umax], s_py[umax]);
    return m == norm;
  }
  return m > p.Rc;
}

// FAST = the configuration BASELINE.json's metric is quoted on (the toy YAML's
// State flags: one-hot action + type-2 piggybacked histogram, my_step, f32
// outputs, no channel-obs / arrival / PRR / PF side outputs).  It is the same
// code with the optional branches removed at compile time; every other config
// runs FAST=false.
template <int VPL, bool FAST>
__global__ __launch_bounds__(Geo<VPL>::THREADS, (VPL == 1 ? DIRAL_MINWAVES : 4)) void step_kernel(const StepParams p) {
  using G = Geo<VPL>;
  constexpr int NPAD = G::NPAD, WAVES = G::WAVES, CC = G::CC, NCH = G::NCH;
  extern __shared__ __align__(16) unsigned char smem[];
  const LdsLayout lay = lds_layout(NPAD, p.A, p.K, VPL, WAVES);
  double* s_px = reinterpret_cast<double*>(smem + lay.px);
  double* s_py = reinterpret_cast<double*>(smem + lay.py);
  double* s_npx = reinterpret_cast<double*>(smem + lay.npx);
  double* s_vel = reinterpret_cast<double*>(smem + lay.vel);
  double* s_rv = reinterpret_cast<double*>(smem + lay.rv);
  double* s_rtx = reinterpret_cast<double*>(smem + lay.rtx);
  double* s_rew = reinterpret_cast<double*>(smem + lay.rew);
  double* s_edges = reinterpret_cast<double*>(smem + lay.edges);
  double* s_red = reinterpret_cast<double*>(smem + lay.red);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem + lay.mask);
  int* s_act = reinterpret_cast<int*>(smem + lay.act);
  int* s_inr = reinterpret_cast<int*>(smem + lay.inr);
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + lay.hist);
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(smem + lay.cnt);
  unsigned char* s_mtab = smem + lay.mtab;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int N = p.N, A = p.A, K = p.K, NV = p.NV;
  const int KP = K | 1;                       // odd histogram row stride: conflict-free LDS
  const int mode = FAST ? (int)DIRAL_STEP_MY_STEP : p.mode;
  const bool do_step = FAST || mode != kModeObserve;
  const bool piggy = FAST || (p.flags & DIRAL_F_ADD_POSDIST_PIGGY);
  // type-1 histograms (network.py:432-471) are built by posdist_kernel.hpp after this launch
  const bool want_hist = FAST || (piggy && p.posdist_type == 2 && p.state_out != nullptr && p.off_hist >= 0);
  const bool track_la = !FAST && (p.flags & DIRAL_F_TRACK_ARRIVAL) && p.la != nullptr;
  const bool want_prr = !FAST && do_step && (mode == DIRAL_STEP_MY_STEP_CH ||
                                              (mode == DIRAL_STEP_MY_STEP && (p.flags & DIRAL_F_TRACK_PRR)));
  const bool mobile = FAST || (p.flags & DIRAL_F_MOBILITY);
  const bool use_pf = !FAST && (p.flags & DIRAL_F_PROPORTIONAL_FAIR);
  const int out_f64 = FAST ? 0 : p.out_f64;
  const size_t bN = (size_t)b * N;
  const size_t bR = (size_t)b * p.NR;   // table row base (padded rows)

  DIRAL_STAMP(0);
  // ---- prefetch (N <= 64): this wave's 16 table columns, before any compute --
  unsigned int pre_w[VPL == 1 ? 16 : 1];
  double pre_x[VPL == 1 ? 16 : 1];
  if constexpr (VPL == 1 && DIRAL_PREFETCH) {
    if (piggy) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int k = wave * 16 + c;
        const bool ok = (k < N) && (lane < N);
        const size_t idx = (bR + (ok ? k : 0)) * NV + (ok ? lane : 0);
        const unsigned int w = p.tkey[idx];
        const double x = p.tx[idx];
        pre_w[c] = ok ? w : 0u;
        pre_x[c] = ok ? x : 0.0;
      }
    }
  }

  // ---- P0: stage per-vehicle state, compute the post-move position --------
  for (int u = tid; u < NPAD; u += G::THREADS) {
    int a = -1;
    double x = 0.0, y = 0.0, v = 0.0, nx = 0.0;
    if (u < N) {
      a = p.actions[bN + u];
      if (a < 0 || a >= A) { atomicOr(p.err, kErrAction); a = -1; }
      x = p.pos_x[bN + u];
      y = p.pos_y[bN + u];
      v = p.vel[bN + u];
      nx = x;
      if (do_step && mobile) {
        if (!FAST && p.trace) {                                    // replay branch, network.py:194-199
          long long tt = (p.t + (p.t_dev ? *p.t_dev : 0ll)) % p.trace_len;
          if (tt < 0) tt += p.trace_len;
          const size_t base = p.trace_per_env ? (size_t)b * p.trace_len : 0;
          nx = p.trace[(base + (size_t)tt) * N + u];
        } else {
          nx = py_mod_pos(x + v + p.L, p.L);                        // network.py:203
        }
      }
    }
    s_act[u] = a; s_px[u] = x; s_py[u] = y; s_vel[u] = v; s_npx[u] = nx;
    s_cnt[u] = 0u; s_rtx[u] = 1.0; s_rew[u] = 0.0; s_inr[u] = 0;
  }
  for (int j = tid; j < KP * NPAD; j += G::THREADS) s_hist[j] = 0u;
  for (int j = tid; j <= K; j += G::THREADS) s_edges[j] = p.edges[j];
  __syncthreads();
  DIRAL_STAMP(1);

  // per-lane copies of this lane's viewers
  int myact[VPL];
  double mypx[VPL], mypy[VPL], mynpx[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int u = lane + 64 * j;
    myact[j] = s_act[u]; mypx[j] = s_px[u]; mypy[j] = s_py[u]; mynpx[j] = s_npx[u];
  }

  // ---- P1: per resource: tx set, closest in-range tx per viewer, rewards ----
  if (do_step) {
    for (int i = wave; i < A; i += WAVES) {
      unsigned long long mk[VPL];
      int c = 0;
#pragma unroll
      for (int j = 0; j < VPL; ++j) { mk[j] = __ballot(myact[j] == i); c += __popcll(mk[j]); }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) s_mask[i * VPL + j] = mk[j];
      }
      double best[VPL];
      int bid[VPL];
#pragma unroll
      for (int j = 0; j < VPL; ++j) { best[j] = 100000.0; bid[j] = -1; }   // network.py:385-386
      if (c > 0) {
        // Network.find_closest_tx (network.py:378-398), ascending tx id, strict <
#pragma unroll
        for (int jt = 0; jt < VPL; ++jt) {
          unsigned long long m = mk[jt];
          while (m) {
            const int w = jt * 64 + __builtin_ctzll(m);
            m &= m - 1;
            const double xw = s_px[w], yw = s_py[w];
            int n_in = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const int u = lane + 64 * j;
              const double d = dist2d(xw, yw, mypx[j], mypy[j]);
              const bool inr = d < p.Rc;
              if (inr && d < best[j]) { best[j] = d; bid[j] = w; }
              if (!FAST) {
                const bool rx = (u < N) && (myact[j] != i);
                if (track_la && rx && !inr) p.la[(bN + w) * N + u] = -1;        // network.py:394
                if (want_prr && c > 1) n_in += __popcll(__ballot(rx && inr));   // test_env.py:395-397
              }
            }
            if (!FAST && want_prr && c > 1 && lane == 0) s_inr[w] = n_in;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const int u = lane + 64 * j;
        const bool is_tx = (myact[j] == i);
        const bool got = (!is_tx) && (bid[j] >= 0) && (u < N) && (c > 0);
        s_mtab[i * NPAD + u] = (unsigned char)(got ? bid[j] : u);
        if (!FAST && u < N) {
          // channel observation of the reference step (`obs[user][i]`)
          double ob = 0.0;
          if (!is_tx && c > 0) {
            if (mode == DIRAL_STEP_MY_STEP && p.state_type == 2) ob = best[j];    // test_env.py:240
            else ob = 1.0;                                                      // :228, :306, :432
          }
          if (p.chobs_out) store_out(p.chobs_out, (bN + u) * A + i, ob, out_f64);
          if (p.state_out && p.off_chobs >= 0)
            store_out(p.state_out, (bN + u) * p.S + p.off_chobs + i, ob, out_f64);
          if (track_la && mode == DIRAL_STEP_MY_STEP_CH && got)
            p.la[(bN + bid[j]) * N + u] = (int32_t)(p.t + (p.t_dev ? *p.t_dev : 0ll));   // test_env.py:436
        }
      }
      if (!FAST && want_prr && c > 1) {
        // received[tx] = #rx whose nearest tx is tx (test_env.py:398-400)
        wave_lds_order();
#pragma unroll
        for (int jt = 0; jt < VPL; ++jt) {
          unsigned long long m = mk[jt];
          while (m) {
            const int w = jt * 64 + __builtin_ctzll(m);
            m &= m - 1;
            int n_rec = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const int u = lane + 64 * j;
              const bool rx = (u < N) && (myact[j] != i);
              n_rec += __popcll(__ballot(rx && bid[j] == w));
            }
            if (lane == 0) {
              const int n_in = s_inr[w];
              s_rtx[w] = n_in > 0 ? (double)n_rec / (double)n_in : 1.0;         // test_env.py:402-405
            }
          }
        }
      }
      if (mode == DIRAL_STEP_MY_STEP && c > 1) {
        // test_env.py:163-199, one value per resource
        double rw = 0.0;
        const int rd = p.reward_design;
        if (rd == 1) {
          const int w = reward_weight<VPL>(p, mk, s_px, s_py);
          const double R = (double)w / (double)c;
          rw = -1.0 * (1.0 - R);
        } else if (rd == 2) {
          if (c == 2) rw = 2.0 * (double)reward_weight<VPL>(p, mk, s_px, s_py) - (double)c;
          else rw = 0.0 - (double)c;
        } else if (rd == 3) {
          const double R = 1.0 / (double)c;
          rw = -1.0 * exp(1.0 - R);
        } else if (rd == 4) {
          rw = 1.0 / (double)c;
        } else {
          if (c == 2) rw = (reward_weight<VPL>(p, mk, s_px, s_py) == 1) ? 0.0 : -1.0;
          else rw = -1.0;
        }
        if (lane == 0) s_rv[i] = rw;
      }
    }
  }
  DIRAL_STAMP(2);
  __syncthreads();
  DIRAL_STAMP(3);

  // ---- P2: reward per transmitter -----------------------------------------
  if (do_step) {
    for (int u = tid; u < NPAD; u += G::THREADS) {
      double r = 0.0, prr = 0.0;
      int sole = 0, coll = 0;
      const int a = s_act[u];
      if (u < N && a >= 0) {
        const int c = popc_masks(s_mask + a * VPL, VPL);
        if (mode == DIRAL_STEP_MY_STEP) {                                     // test_env.py:211-222
          if (c > 1) {
            r = s_rv[a];
            if (use_pf) {
              const int pc = p.pf[bN + u];
              if (pc > p.pf_threshold) r = p.pf_penalty;
              p.pf[bN + u] = pc + 1;
            }
            coll = 1; prr = s_rtx[u];
          } else {
            r = 1.0;
            if (use_pf) p.pf[bN + u] = 0;
            sole = 1; prr = 1.0;
          }
        } else if (mode == DIRAL_STEP_MY_STEP_CH) {                           // test_env.py:411-429
          const int rd = p.reward_design;
          if (c > 1) {
            const double R = s_rtx[u];
            if (rd == 3) r = 1.0 - exp(1.0 - R);
            else if (rd == 4) r = -1.0 * exp(1.0 - R);
            else if (rd == 2) r = -1.0 * (1.0 - R);
            coll = 1; prr = R;
          } else {
            if (rd == 3) r = 1.0;
            else if (rd == 4) r = exp(1.0);
            else if (rd == 2) r = 1.0;
            sole = 1; prr = 1.0;
          }
        } else {                                                              // test_env.py:297-301, 319-349
          if (c == 1) { r = 1.0; sole = 1; }
          else {
            int n = 1;
            double dlast = 0.0;
            for (int jt = 0; jt < VPL; ++jt) {
              unsigned long long m = s_mask[a * VPL + jt];
              while (m) {
                const int o = jt * 64 + __builtin_ctzll(m);
                m &= m - 1;
                if (o == u) continue;
                const double d = dist2d(s_px[u], s_py[u], s_px[o], s_py[o]);
                if (d < 2.0 * p.Rc) { if (n == 1) dlast = d; ++n; }           // network.py:122-133
              }
            }
            if (n == 1) r = 1.0;
            else if (n == 2) r = ((dlast / 1.0) > p.Rc * 2.0) ? 0.0 : -2.0;    // network.py:135-157
            else r = -(double)n;
            coll = 1;
          }
        }
        s_rew[u] = r;
        if (p.rew_out) store_out(p.rew_out, bN + u, r, out_f64);
      }
      // deterministic per-wave reductions for the metric accumulators
      double vr = r, vp = want_prr ? prr : 0.0;
      int vs = sole, vc = coll;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        vr += __shfl_down(vr, off);
        if (!FAST) vp += __shfl_down(vp, off);
        vs += __shfl_down(vs, off);
        vc += __shfl_down(vc, off);
      }
      if (lane == 0) {
        const int slot = (u >> 6);
        s_red[slot * 4 + 0] = vr; s_red[slot * 4 + 1] = vp;
        s_red[slot * 4 + 2] = (double)vs; s_red[slot * 4 + 3] = (double)vc;
      }
    }
  } else {
    for (int u = tid; u < N; u += G::THREADS) s_rew[u] = p.rew_in ? p.rew_in[bN + u] : 0.0;
  }

  DIRAL_STAMP(4);
  // ---- P3: neighbour-table stamp + gossip merge + observation histogram ----
  if (piggy && (do_step || want_hist)) {
    unsigned int* scratch = reinterpret_cast<unsigned int*>(smem + lay.scratch) + wave * 1024;
    unsigned int mycnt[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) mycnt[j] = 0u;
    const double inv_w = p.hist_inv_width;

    // load + Vehicle.periodic_update (vehicle.py:56-70), branch-free
    bool seq_ovf = false;
    auto load_stamp = [&](int k, int j, unsigned int pre) -> unsigned int {
      const int u = lane + 64 * j;
      unsigned int w = pre;
      if (VPL > 1 || !DIRAL_PREFETCH) w = (k < N && u < N) ? p.tkey[(bR + k) * NV + u] : 0u;
      if (do_step) {
        const bool own = (u == k) && (u < N);
        const unsigned int seq = (w >> 8) + (own ? 1u : 0u);
        const unsigned int a0 = w & 255u;
        const unsigned int age = own ? 0u : (a0 + (a0 < 255u ? 1u : 0u));
        seq_ovf = seq_ovf || (own && seq >= (1u << 24) - 1u);
        w = (seq << 8) | age;
      }
      return w;
    };

    // finalize one column: xpos follows the winning sequence number, age resets on
    // change, then the viewer-side histogram contribution of the entry
    auto finalize = [&](int k, const unsigned int* kf_j, const unsigned int* w_j, const double* xo_j) {
      const double pxk = s_px[k], pyk = s_py[k];
      double xn[VPL];
      unsigned int wn[VPL];
      bool changed[VPL];
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const int u = lane + 64 * j;
        const unsigned int kf = kf_j[j], w = w_j[j];
        const unsigned int src = kf & 255u;
        const bool upd = ((kf ^ w) >> 8) != 0u;
        double xo = xo_j[j];                                   // loaded by the caller, all columns at once
        if (do_step && u == k) xo = pxk;                       // own stamp (vehicle.py:63)
        double xg = xo;
        if constexpr (VPL == 1) {
          const int lo = __builtin_amdgcn_ds_bpermute((int)src << 2, __double2loint(xo));
          const int hi = __builtin_amdgcn_ds_bpermute((int)src << 2, __double2hiint(xo));
          if (upd) xg = __hiloint2double(hi, lo);
        }
        xn[j] = xg;
        wn[j] = upd ? (kf & ~255u) : w;
        changed[j] = upd;
      }
      if constexpr (VPL > 1) {
        // N > 64: the source viewer may sit in another lane slot - stage the column's
        // xpos in the wave's LDS scratch (free after the merge) and gather from there;
        // un-branched (a lane that did not update re-reads its own value)
        double* sx = reinterpret_cast<double*>(scratch);
#pragma unroll
        for (int j = 0; j < VPL; ++j) sx[lane + 64 * j] = xn[j];
        wave_lds_order();
        double xg[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) xg[j] = sx[changed[j] ? (int)(kf_j[j] & 255u) : lane + 64 * j];
        wave_lds_order();
#pragma unroll
        for (int j = 0; j < VPL; ++j) xn[j] = xg[j];
      }
      if (do_step) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const int u = lane + 64 * j;
          if (u < N) {
            p.tkey[(bR + k) * NV + u] = wn[j];
            if (changed[j] || u == k) p.tx[(bR + k) * NV + u] = xn[j];
          }
        }
      }
      if (want_hist) {
        // Network.dist_piggy + get_positional_dist_2_piggy (network.py:538-558, 473-513)
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const int u = lane + 64 * j;
          const int age = (int)(wn[j] & 255u);
          if (u < N && u != k && age < p.age_limit) {
            const double x1 = xn[j];
            const double y1 = ((wn[j] >> 8) > 0u) ? pyk : 0.0;
            // (N > 64 re-reads its own position from LDS: keeping 3*VPL doubles live across the
            // merge spills under the 128-VGPR cap of a 1024-thread workgroup)
            const double x2 = (VPL == 1) ? mynpx[j] : s_npx[u], y2 = (VPL == 1) ? mypy[j] : s_py[u];
            const double d = dist2d(x1, y1, x2, y2);
            if (d < p.Rb) {
              const double v = (x1 - x2 > 0.0) ? d : -d;
              const int bin = hist_bin(v, -p.Rb, inv_w, K, s_edges);
              atomicAdd(&s_hist[u * KP + bin], 1u);
              mycnt[j] += 1u;
            }
          }
        }
      }
    };

    // Resources with at least one transmitter, as wave-uniform bit words (computed
    // once: no per-step LDS round trip for the mask), visited in ascending order with
    // the NEXT resource's gather sources already in flight.
    unsigned long long act_w[4] = {0ull, 0ull, 0ull, 0ull};
    if (VPL > 1 && do_step) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = q * 64 + lane;
        unsigned long long any = 0ull;
        if (q * 64 < A && i < A) {
#pragma unroll
          for (int j = 0; j < VPL; ++j) any |= s_mask[i * VPL + j];
        }
        act_w[q] = __ballot(any != 0ull);
      }
    }
    auto for_each_active_resource = [&](auto&& body) {
      int m_next[VPL];
      bool have_next = false;
      const int nwords = (A + 63) >> 6;
#pragma unroll 1
      for (int q = 0; q < nwords; ++q) {                       // rolled: one copy of the body
        unsigned long long w = q == 0 ? act_w[0] : (q == 1 ? act_w[1] : (q == 2 ? act_w[2] : act_w[3]));
        while (w) {
          const int i = q * 64 + __builtin_ctzll(w);
          w &= w - 1;
          int m[VPL];
#pragma unroll
          for (int j = 0; j < VPL; ++j) m[j] = have_next ? m_next[j] : (int)s_mtab[i * NPAD + lane + 64 * j];
          have_next = (w != 0ull);                             // prefetch within the word only
          if (have_next) {
            const int inext = q * 64 + __builtin_ctzll(w);
#pragma unroll
            for (int j = 0; j < VPL; ++j) m_next[j] = s_mtab[inext * NPAD + lane + 64 * j];
          }
          body(m);
        }
      }
    };

    // 32-bit merge of NC columns held in LDS scratch (N > 64): Vehicle.received_update
    // for every (resource, rx) in reference order; a wave's own LDS queue is in order
    auto merge32_wide = [&](unsigned int* key, auto nc_tag) {
      constexpr int NC = decltype(nc_tag)::value;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < VPL; ++j) scratch[c * NPAD + lane + 64 * j] = key[c * VPL + j];
      wave_lds_order();
      for_each_active_resource([&](const int (&m)[VPL]) {
        unsigned int v[NC * VPL];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int j = 0; j < VPL; ++j) v[c * VPL + j] = scratch[c * NPAD + m[j]];
        // tx entries are not written in resource i (a tx never merges on its own
        // resource), so all gathers of step i may precede all writes
        wave_lds_order();
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            const unsigned int nk = max(key[c * VPL + j], v[c * VPL + j]);
            // unconditional: one ds_write is cheaper to ISSUE than compare + exec-mask + write
            scratch[c * NPAD + lane + 64 * j] = nk;
            key[c * VPL + j] = nk;
          }
        wave_lds_order();
      });
    };

    if constexpr (VPL == 1) {
      const int kbase = wave * 16;
      if (kbase < N) {
        unsigned int w1[16], key[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          w1[c] = load_stamp(kbase + c, 0, DIRAL_PREFETCH ? pre_w[c] : 0u);
          key[c] = (w1[c] & ~255u) | (unsigned int)lane;
        }
        if (do_step) {
          for (int i = 0; i < A; ++i) {                          // one ds_bpermute + max per column
            if (uniform_u64(s_mask[i]) == 0ull) continue;
            const int m4 = (int)s_mtab[i * NPAD + lane] << 2;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              const unsigned int v = (unsigned int)__builtin_amdgcn_ds_bpermute(m4, (int)key[c]);
              key[c] = max(key[c], v);
            }
          }
        }
        DIRAL_STAMP(5);
        double xo1[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (DIRAL_PREFETCH) xo1[c] = pre_x[c];
          else xo1[c] = (kbase + c < N && lane < N) ? p.tx[(bR + kbase + c) * NV + lane] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (kbase + c < N) finalize(kbase + c, &key[c], &w1[c], &xo1[c]);
      }
    } else {
      // N > 64.  CC columns per pass (16 key registers); packed as CC/2 column PAIRS of 16-bit keys
      // (rank << 8) | source, rank = 255 - lag, merged with v_pk_max_u16 - half the
      // LDS traffic and VALU work of the 32-bit merge.  Exact while no entry of the
      // pass has lag >= 255 with seq != 0 (see step_fast64.hpp for the argument);
      // otherwise the pass takes the 32-bit merge.
      constexpr int PC = CC, HP = CC / 2;
      for (int pch = 0; pch < 16 / PC; ++pch) {
        const int kbase = wave * 16 + pch * PC;
        if (kbase >= N) break;
        unsigned int w1[PC * VPL], key[PC * VPL];
#pragma unroll
        for (int c = 0; c < PC; ++c)
#pragma unroll
          for (int j = 0; j < VPL; ++j) w1[c * VPL + j] = load_stamp(kbase + c, j, 0u);
        // (key[] is built inside each branch so it is not live across the packed merge)
        auto init_key = [&]() {
#pragma unroll
          for (int c = 0; c < PC; ++c)
#pragma unroll
            for (int j = 0; j < VPL; ++j)
              key[c * VPL + j] = (w1[c * VPL + j] & ~255u) | (unsigned int)(lane + 64 * j);
        };
        if (!do_step) init_key();
        if (do_step) {
          typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
          unsigned int kp[HP * VPL];
          unsigned int tk_own[PC];
          bool bad = false;
#pragma unroll
          for (int c = 0; c < PC; ++c) {
            const int k = kbase + c;
            // the subject's own fresh sequence number: entry (viewer k, subject k)
            unsigned int tko = 0u;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const unsigned int cand = (unsigned int)__builtin_amdgcn_readlane((int)(w1[c * VPL + j] >> 8), k & 63);
              tko = ((k >> 6) == j) ? cand : tko;
            }
            tk_own[c] = tko;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const unsigned int seq = w1[c * VPL + j] >> 8;
              const unsigned int lag = tko - seq;
              bad = bad || (lag >= 255u && seq != 0u);
              const unsigned int rank = lag < 255u ? 255u - lag : 0u;
              const unsigned int k16 = (rank << 8) | (unsigned int)(lane + 64 * j);
              if ((c & 1) == 0) kp[(c >> 1) * VPL + j] = k16;
              else kp[(c >> 1) * VPL + j] |= k16 << 16;
            }
          }
          const bool packed_ok = (DIRAL_PACKED_WIDE != 0) && (__ballot(bad) == 0ull);
          if (packed_ok) {
#pragma unroll
            for (int c2 = 0; c2 < HP; ++c2)
#pragma unroll
              for (int j = 0; j < VPL; ++j) scratch[c2 * NPAD + lane + 64 * j] = kp[c2 * VPL + j];
            wave_lds_order();
            for_each_active_resource([&](const int (&m)[VPL]) {
              unsigned int v[HP * VPL];
#pragma unroll
              for (int c2 = 0; c2 < HP; ++c2)
#pragma unroll
                for (int j = 0; j < VPL; ++j) v[c2 * VPL + j] = scratch[c2 * NPAD + m[j]];
              wave_lds_order();
#pragma unroll
              for (int c2 = 0; c2 < HP; ++c2)
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                  const unsigned int old = kp[c2 * VPL + j];
                  const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, old),
                                                            __builtin_bit_cast(u16x2, v[c2 * VPL + j]));
                  const unsigned int nk = __builtin_bit_cast(unsigned int, r);
                  scratch[c2 * NPAD + lane + 64 * j] = nk;
                  kp[c2 * VPL + j] = nk;
                }
              wave_lds_order();
            });
            // back to (seq << 8) | source; rank 0 never results from an update
#pragma unroll
            for (int c = 0; c < PC; ++c)
#pragma unroll
              for (int j = 0; j < VPL; ++j) {
                const unsigned int k16 = (kp[(c >> 1) * VPL + j] >> (16 * (c & 1))) & 0xffffu;
                const unsigned int rank = k16 >> 8, src = k16 & 255u;
                const unsigned int seqf = tk_own[c] - 255u + rank;
                const unsigned int msk = 0u - (unsigned int)(rank != 0u);
                const unsigned int own_key = (w1[c * VPL + j] & ~255u) | (unsigned int)(lane + 64 * j);
                key[c * VPL + j] = (((seqf << 8) | src) & msk) | (own_key & ~msk);
              }
          } else {
            init_key();
            merge32_wide(&key[0], std::integral_constant<int, CC>{});
          }
        }
        DIRAL_STAMP(5);
        // every xpos load of the pass is issued before the first column is finalized
        // (the LDS-ordering barriers inside finalize() would otherwise pin each
        // column's loads behind the previous column: one HBM round trip per column)
        double xo_all[PC * VPL];
#pragma unroll
        for (int c = 0; c < PC; ++c)
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            const int u = lane + 64 * j;
            xo_all[c * VPL + j] = (kbase + c < N && u < N) ? p.tx[(bR + kbase + c) * NV + u] : 0.0;
          }
#pragma unroll
        for (int c = 0; c < PC; ++c)
          if (kbase + c < N) finalize(kbase + c, &key[c * VPL], &w1[c * VPL], &xo_all[c * VPL]);
      }
    }
    if (seq_ovf) atomicOr(p.err, kErrSeq);
    if (want_hist) {
#pragma unroll
      for (int j = 0; j < VPL; ++j)
        if (mycnt[j]) atomicAdd(&s_cnt[lane + 64 * j], mycnt[j]);
    }
  }
  DIRAL_STAMP(6);
  __syncthreads();
  DIRAL_STAMP(7);

  // ---- P4: write-back: positions, state vectors, done flag, metrics --------
  if (do_step) {
    for (int u = tid; u < N; u += G::THREADS) p.pos_x[bN + u] = s_npx[u];
    if (tid == 0) {
      if (p.done_out) p.done_out[b] = (uint8_t)(((p.t + (p.t_dev ? *p.t_dev : 0ll)) % p.episode_interval) == p.episode_interval - 1);
      double sr = 0.0, sp = 0.0, ss = 0.0, sc = 0.0;
      for (int w = 0; w < VPL; ++w) {
        sr += s_red[w * 4 + 0]; sp += s_red[w * 4 + 1]; ss += s_red[w * 4 + 2]; sc += s_red[w * 4 + 3];
      }
      double* mt = p.metrics + (size_t)b * DIRAL_M_COLUMNS;
      mt[DIRAL_M_SLOTS] += 1.0;
      mt[DIRAL_M_SUM_REWARD] += sr;
      mt[DIRAL_M_TX_SOLE] += ss;
      mt[DIRAL_M_TX_COLLIDED] += sc;
      if (want_prr) { mt[DIRAL_M_PRR_SUM] += sp; mt[DIRAL_M_PRR_CNT] += ss + sc; }
    }
  }
  if (FAST) {
    // state = [one-hot(action) (A) | histogram (K)], float32, S = A + K
    const int S = A + K;
    float* out = reinterpret_cast<float*>(p.state_out) + bN * S;
    if (((A | K) & 3) == 0) {
      // 16-byte stores: rows are S*4 bytes, S % 4 == 0
      const int q_per_row = S >> 2, total = N * q_per_row;
      for (int q = tid; q < total; q += G::THREADS) {
        const int u = q / q_per_row, s0 = (q - u * q_per_row) << 2;
        float4 v;
        if (s0 < A) {
          const int a = s_act[u] - s0;
          v = make_float4(a == 0 ? 1.f : 0.f, a == 1 ? 1.f : 0.f, a == 2 ? 1.f : 0.f, a == 3 ? 1.f : 0.f);
        } else {
          const unsigned int n = s_cnt[u];
          const unsigned int* h = s_hist + u * KP + (s0 - A);
          // (float)((double)h/(double)n) == correctly rounded float division for
          // integers h <= n <= 255 (the quotient is never within 2^-53 of a
          // float midpoint), so the f32 division is exact w.r.t. network.py:501
          const float fn = (float)n;
          v = n ? make_float4(__fdiv_rn((float)h[0], fn), __fdiv_rn((float)h[1], fn),
                              __fdiv_rn((float)h[2], fn), __fdiv_rn((float)h[3], fn))
                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        reinterpret_cast<float4*>(out)[q] = v;
      }
    } else {
      for (int e = tid; e < N * S; e += G::THREADS) {
        const int u = e / S, s = e - u * S;
        float val;
        if (s < A) val = (s_act[u] == s) ? 1.f : 0.f;
        else {
          const unsigned int n = s_cnt[u];
          val = n ? __fdiv_rn((float)s_hist[u * KP + (s - A)], (float)n) : 0.f;
        }
        out[e] = val;
      }
    }
  } else if (p.state_out) {
    const int S = p.S;
    const int total = N * S;
    int e = tid;
    int u = e / S, s = e - u * S;
    const int du = G::THREADS / S, ds = G::THREADS - du * S;
    while (e < total) {
      double val = 0.0;
      bool write = true;
      if (p.off_act >= 0 && s >= p.off_act && s < p.off_act + ((p.flags & DIRAL_F_ACTION_REAL) ? 1 : A)) {
        val = (p.flags & DIRAL_F_ACTION_REAL) ? (double)s_act[u]
                                               : ((s_act[u] == s - p.off_act) ? 1.0 : 0.0);   // test_env.py:585-595
      } else if (p.off_chobs >= 0 && s >= p.off_chobs && s < p.off_chobs + A) {
        if (do_step) write = false;   // written in P1
        else val = p.chobs_in ? p.chobs_in[(bN + u) * A + (s - p.off_chobs)] : 0.0;
      } else if (p.off_hist >= 0 && s >= p.off_hist && s < p.off_hist + K) {
        if (p.posdist_type != 2) write = false;   // written by posdist_kernel
        const unsigned int n = s_cnt[u];
        const unsigned int h = s_hist[u * KP + (s - p.off_hist)];
        val = n ? (double)h / (double)n : 0.0;                                   // network.py:501
      } else if (p.off_posdist >= 0 && s >= p.off_posdist && s < p.off_posdist + N - 1) {
        write = false;                            // written by posdist_kernel
      } else if (s == p.off_rew) {
        val = s_rew[u];
      } else if (s == p.off_idx) {
        val = (double)(u + 1);
      } else if (p.off_pos >= 0 && s == p.off_pos) {
        val = s_npx[u] / p.L;                                                    // network.py:403-407
      } else if (p.off_pos >= 0 && s == p.off_pos + 1) {
        val = s_py[u] / p.H;
      } else if (s == p.off_vel) {
        val = s_vel[u];
      } else if (p.off_fp >= 0 && s == p.off_fp) {
        val = p.episode;
      } else if (p.off_fp >= 0 && s == p.off_fp + 1) {
        val = p.eps;
      }
      if (write) store_out(p.state_out, (size_t)bN * S + e, val, out_f64);
      e += G::THREADS; u += du; s += ds;
      if (s >= S) { s -= S; u += 1; }
    }
  }
}

}  // namespace diral
