// observe_kernel.hpp - TestEnv.obtain_state (test_env.py:527-583) as a launch of its own, any
// configuration this build accepts (N <= 256, any State flags).
//
// The step kernels build the state vector of the slot they just advanced (`step()`, and the
// speculative state of `my_step*`).  A stand-alone obtain_state - called with actions, channel
// observation or rewards other than what the step returned (diral_env_observe) - changes nothing
// in the env: it reads the tables and positions as they are and writes [B][N][S].  Until round 3
// that call ran the general step kernel in an observe-only mode (C2: ~230 us); this kernel does
// just the finalize + output work:
//   * one workgroup per env, wave w sweeps the subject rows k = w, w + 4, ... of the subject-major
//     table: a row is "what every viewer knows about k", lane = viewer (+ 64 j), one coalesced
//     4-byte load per lane;
//   * the xpos of an entry that lags its subject by at most 7 stamps comes from the subject's
//     ring row (the env's rows and the subjects' own sequence numbers are staged in LDS
//     first, so the sweep has one global load per entry, eight rows in flight per lane), older ones
//     from the per-entry plane - the plane does not have to be materialised first;
//   * Network.dist_piggy + get_positional_dist_2_piggy (network.py:538-558, 473-513) per entry,
//     histogram rows in LDS, then the generic state writer of rich_out.hpp with the caller's
//     `obs` / `rewards` arrays as the channel-observation and reward sections.
// The secondary observation columns (sorted true distances, type-1 histogram) stay with
// posdist_kernel.hpp, which diral_env_observe launches right after this one, as after a step.
#pragma once
#include "common.hpp"
#include "rich_out.hpp"
#include "step_fast64.hpp"
#include "step_kernel.hpp"

namespace diral {

struct ObserveParams {
  int N, A, K, NV, NR;
  uint32_t flags;
  int age_limit;
  int want_hist;                 // the state carries the type-2 piggybacked histogram
  double L, Rb, inv_w;
  const int32_t* actions;        // [B][N] the `acts` argument
  const double* chobs_in;        // [B][N][A] the `obs` argument (f64) or null: zeros
  const double* rew_in;          // [B][N] the `rewards` argument (f64) or null: zeros
  const double* pos_x;
  const double* pos_y;
  const double* vel;
  const uint32_t* tkey;
  const double* tx;
  const double* ring;            // [B][NR][8] or null: every xpos from the plane
  const uint32_t* tcode;         // the packed table of step_fast64 (N <= 64, step_fast64.hpp) or null: sequence numbers and
  const uint32_t* tage;          // ages from `tkey`.  Set together with `ring`: codes, ages, own sequence numbers;
  const uint32_t* tseq;          // `tkey` / `tx` then only answer for the code-0 entries (never heard, or older than 7)
  const double* edges;
  const double* inv_tab;         // [256] 1.0 / n (0 for n = 0), as in step_fast64.hpp
  uint32_t* err;
  void* state_out;
};

struct ObserveLds {
  uint32_t px, py, edges, ring, act, cnt, tk, hist, total;
};
__host__ __device__ inline ObserveLds observe_lds_layout(int N, int K) {
  const uint32_t npad = (uint32_t)align_up((uint32_t)N, 64);
  ObserveLds l;
  uint32_t o = 0;
  l.px = o;    o += 8u * npad;
  l.py = o;    o += 8u * npad;
  l.edges = o; o += 8u * (K + 2);
  l.ring = o;  o += 8u * 8u * npad;                  // the env's ring rows [subject][8]
  l.act = o;   o += 4u * npad;
  l.cnt = o;   o += 4u * npad;
  l.tk = o;    o += 4u * npad;                       // the subjects' own sequence numbers
  l.hist = o;  o += 4u * (uint32_t)(K | 1) * npad;   // [viewer][K | 1]: odd row stride
  l.total = align_up(o, 16);
  return l;
}

constexpr int kObserveThreads = 256;

template <bool FLAT, bool OUT64>
__global__ __launch_bounds__(kObserveThreads) void observe_kernel(const ObserveParams p, const RichParams r) {
  extern __shared__ __align__(16) unsigned char smem[];
  const ObserveLds lay = observe_lds_layout(p.N, p.K);
  double* s_px = reinterpret_cast<double*>(smem + lay.px);
  double* s_py = reinterpret_cast<double*>(smem + lay.py);
  double* s_edges = reinterpret_cast<double*>(smem + lay.edges);
  double* s_ring = reinterpret_cast<double*>(smem + lay.ring);
  unsigned int* s_tk = reinterpret_cast<unsigned int*>(smem + lay.tk);
  int* s_act = reinterpret_cast<int*>(smem + lay.act);
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(smem + lay.cnt);
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + lay.hist);

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, A = p.A, K = p.K, NV = p.NV;
  const int KP = K | 1;
  const size_t bN = (size_t)b * N, bR = (size_t)b * p.NR;

  for (int u = tid; u < N; u += kObserveThreads) {
    int a = p.actions[bN + u];
    if (a < 0 || a >= A) { atomicOr(p.err, kErrAction); a = -1; }
    s_act[u] = a;
    s_px[u] = p.pos_x[bN + u];
    s_py[u] = FLAT ? 0.0 : p.pos_y[bN + u];
    s_cnt[u] = 0u;
  }
  const bool use_ring = p.ring != nullptr;
  if (p.want_hist) {
    for (int j = tid; j < KP * N; j += kObserveThreads) s_hist[j] = 0u;
    if (tid <= K + 1) s_edges[tid] = p.edges[tid < K ? tid : K];
    // the env's ring rows (coalesced) and the subjects' own sequence numbers (the table's diagonal) into LDS:
    // the sweep below then has ONE global load per entry - the table word - and nothing that depends on it
    // except LDS reads (the xpos of the rare entry older than the ring reaches comes from the plane)
    if (use_ring)
      for (int j = tid; j < 8 * N; j += kObserveThreads) s_ring[j] = p.ring[bR * 8 + j];
    for (int k = tid; k < N; k += kObserveThreads) s_tk[k] = p.tseq ? p.tseq[bR + k] : p.tkey[(bR + k) * NV + k] >> 8;
  }
  __syncthreads();

  if (p.want_hist) {
    // Network.dist_piggy + get_positional_dist_2_piggy (network.py:538-558, 473-513): viewer u against its
    // entry about k, own position as it is NOW (obtain_state runs after the move)
    const double inv_w = p.inv_w;
    constexpr int RW = kObserveThreads / 64;         // waves: wave w sweeps rows (or row-quads) w, w + RW, ...
    constexpr int G = 8;                             // rows whose table words a lane has in flight together
    // one entry: viewer u (position mx, my) against its entry about k - age, xpos xg, heard at all
    auto entry = [&](int u, int k, double mx, double my, unsigned int age, bool heard, double xg, unsigned int* hrow,
                     unsigned int& mycnt) {
      double d, v;
      if constexpr (FLAT) {
        v = xg - mx;                                                  // all y == 0: x1 - x2 IS d * sign, d = |v| (a square
        d = __builtin_fabs(v);                                        // that underflows: the edge branch, see step_fast64.hpp)
      } else {
        d = fast_dist<false>(xg, heard ? s_py[k] : 0.0, mx, my);      // ypos: the subject's lane once heard
        v = (xg - mx > 0.0) ? d : -d;
      }
      if (u != k && (int)age < p.age_limit && d < p.Rb) {
        bool unsafe;
        int bin = hist_bin_estimate(v, p.Rb, inv_w, K, unsafe);    // (step_kernel.hpp: the edges are read only near an edge)
        if (unsafe) {
          bin = hist_bin_clamp(bin, K);
          if constexpr (FLAT) {
            if (((unsigned int)__double2hiint(v) & 0x7fffffffu) < 0x20b00000u) {   // |v| below 2^-500 (its square underflows) or 0
              d = dist_general(mx - xg, 0.0);
              v = (v > 0.0) ? d : -d;
            }
          }
          const double e0 = s_edges[bin], e1 = s_edges[bin + 1];
          bin += (v >= e1 ? 1 : 0) - (v < e0 ? 1 : 0);
        }
        atomicAdd(&hrow[bin], 1u);
        mycnt += 1u;
      }
    };
    if (p.tcode) {
      // the packed table: one code word + one age word per row-quad and viewer = four entries
      const size_t bQ = (size_t)b * (p.NR >> 2);
      const int nq = (N + 3) >> 2;
      for (int u = lane; u < N; u += 64) {
        const double mx = s_px[u], my = s_py[u];
        unsigned int mycnt = 0u;
        unsigned int* const hrow = s_hist + u * KP;
#pragma unroll 1
        for (int qb = wave; qb < nq; qb += RW * 4) {
        unsigned int cv[4], av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = qb + RW * i;
          cv[i] = p.tcode[(bQ + (q < nq ? q : 0)) * NV + u];
          av[i] = p.tage[(bQ + (q < nq ? q : 0)) * NV + u];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = qb + RW * i;
          if (q >= nq) break;                                          // wave-uniform
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = 4 * q + c;
            if (k >= N) break;
            const unsigned int rcode = (cv[i] >> (8 * c)) & 255u, age = (av[i] >> (8 * c)) & 255u;
            double xg;
            bool heard = true;
            if (rcode) {
              xg = s_ring[k * 8 + ((s_tk[k] - 8u + (unsigned int)__popc(rcode)) & 7u)];
            } else {                                                   // never heard, or older than the codes reach
              xg = p.tx[(bR + k) * NV + u];
              if constexpr (!FLAT) heard = (p.tkey[(bR + k) * NV + u] >> 8) != 0u;
            }
            entry(u, k, mx, my, age, heard, xg, hrow, mycnt);
          }
        }
        }
        if (mycnt) atomicAdd(&s_cnt[u], mycnt);
      }
    } else {
    for (int u = lane; u < N; u += 64) {
      const double mx = s_px[u], my = s_py[u];
      unsigned int mycnt = 0u;
      unsigned int* const hrow = s_hist + u * KP;
#pragma unroll 1
      for (int k0 = wave; k0 < N; k0 += RW * G) {
        unsigned int wv[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int k = k0 + RW * i;
          wv[i] = p.tkey[(bR + (k < N ? k : k0)) * NV + u];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int k = k0 + RW * i;
          if (k >= N) break;                                          // wave-uniform
          const unsigned int wn = wv[i];
          const unsigned int seq = wn >> 8;
          double xg;
          if (use_ring && s_tk[k] - seq <= 7u) xg = s_ring[k * 8 + (seq & 7u)];
          else xg = p.tx[(bR + k) * NV + u];
          entry(u, k, mx, my, wn & 255u, seq != 0u, xg, hrow, mycnt);
        }
      }
      if (mycnt) atomicAdd(&s_cnt[u], mycnt);
    }
    }
    __syncthreads();
  }

  if (r.plain_state && ((A | K) & 3) == 0) {
    // the toy YAML's state vector [one-hot(action) | histogram]: 16 bytes per lane, consecutive lanes on
    // consecutive pieces of a row (the vectorised writer of step_fast64.hpp's P4; values identical)
    const int S = A + K;
    const int q_per_row = S >> 2, total = N * q_per_row;
    const int du = kObserveThreads / q_per_row, dq = kObserveThreads - du * q_per_row;
    int u = tid / q_per_row, qr = tid - u * q_per_row;
    for (int q = tid; q < total; q += kObserveThreads) {
      const int s0 = qr << 2;
      double v0, v1, v2, v3;
      if (s0 < A) {
        const int a = s_act[u] - s0;
        v0 = a == 0 ? 1.0 : 0.0; v1 = a == 1 ? 1.0 : 0.0; v2 = a == 2 ? 1.0 : 0.0; v3 = a == 3 ? 1.0 : 0.0;
      } else {
        const unsigned int n = s_cnt[u];
        const unsigned int* h = s_hist + u * KP + (s0 - A);
        if constexpr (OUT64) {
          const double dn = (double)n;                               // network.py:501: one IEEE division per bin
          v0 = n ? (double)h[0] / dn : 0.0; v1 = n ? (double)h[1] / dn : 0.0;
          v2 = n ? (double)h[2] / dn : 0.0; v3 = n ? (double)h[3] / dn : 0.0;
        } else {
          // (float)((double)h * fl(1.0 / n)) == (float)((double)h / (double)n) for 0 <= h <= n <= 255 (step_fast64.hpp)
          const double inv = p.inv_tab[n < 256u ? n : 0u];
          v0 = (double)h[0] * inv; v1 = (double)h[1] * inv; v2 = (double)h[2] * inv; v3 = (double)h[3] * inv;
        }
      }
      if constexpr (OUT64) {
        double* out = static_cast<double*>(p.state_out) + bN * S + 4 * (size_t)q;
        stream_store2(out, make_double2(v0, v1));
        stream_store2(out + 2, make_double2(v2, v3));
      } else {
        stream_store4(static_cast<float*>(p.state_out) + bN * S + 4 * (size_t)q, make_float4((float)v0, (float)v1, (float)v2, (float)v3));
      }
      u += du; qr += dq;
      if (qr >= q_per_row) { qr -= q_per_row; u += 1; }
    }
    return;
  }
  const double* const chobs_in = p.chobs_in;
  const double* const rew_in = p.rew_in;
  rich_write_state<OUT64>(
      r, p.flags, N, A, K, p.L, p.state_out, bN, tid, kObserveThreads, [&](int u) { return s_act[u]; },
      [&](int u, int i) { return chobs_in ? chobs_in[(bN + u) * A + i] : 0.0; },
      [&](int u, int bin) {
        const unsigned int n = s_cnt[u];
        return n ? (double)s_hist[u * KP + bin] / (double)n : 0.0;     // network.py:501
      },
      [&](int u) { return rew_in ? rew_in[bN + u] : 0.0; }, [&](int u) { return s_px[u]; },
      [&](int u) { return s_py[u]; }, [&](int u) { return r.vel[bN + u]; });
}

}  // namespace diral
