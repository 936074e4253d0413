// rich_out.hpp - the RICH output tail of the specialised step kernels.
//
// The plain instantiations of step_fast64 / step_wide write exactly the toy YAML's
// state vector [one-hot(action) | type-2 histogram].  The RICH instantiations
// serve everything else that is only an extra OUTPUT of the same slot:
//   * the channel observation `obs` the reference's step functions return
//     (test_env.py:143, 206, 228, 240, 306, 432) into `chobs_out`, so the
//     reference's own call pattern `obs, rews = env.my_step(a, t);
//     env.obtain_state(obs, a, rews)` (main_test.py:144-164) stays on the
//     specialised kernels;
//   * the cheap State flags of TestEnv.obtain_state (test_env.py:527-583):
//     action_index "real", add_channel_obs, add_reward, add_index, add_position,
//     add_velocity, enable_fingerprint, and State.type 1 (channel observation 1
//     instead of the distance, test_env.py:226-240);
//   * proportional fairness (test_env.py:211-222: the pf_counter penalty of my_step).
// Nothing here feeds back into the step: the values are rebuilt in the output
// phase from what the step left in LDS (gather sources, transmitter masks,
// positions), so the hot loops of the plain instantiations are untouched.
#pragma once
#include "common.hpp"

namespace diral {

// second kernel argument of the specialised kernels; only RICH instantiations read it
struct RichParams {
  void* chobs_out;         // [B][N][A] (out dtype) or null
  int S;                   // state_space
  int state_type;          // State.type: 2 = distance to the closest transmitter, 1 = constant 1
  int plain_state;         // the state vector is the plain [one-hot | histogram]: the vectorised writer serves it
  int off_act, off_chobs, off_hist, off_rew, off_idx, off_pos, off_vel, off_fp;   // -1 = absent
  int off_skip, len_skip;  // columns [off_skip, off_skip + len_skip) belong to another launch (posdist_kernel): not written
  double H;                // highway_height (network.py:31): pos_y / H
  double episode, eps;     // fingerprint (test_env.py:577-579)
  const double* vel;       // [B][N] (add_velocity reads it in the output phase)
  const double* pos_y;     // [B][N]
  int32_t* pf;             // [B][N] pf_counter (test_env.py:87-92) or null: proportional fairness off
  int pf_threshold;        // 10  test_env.py:89
  double pf_penalty;       // -10 test_env.py:90
};

template <bool OUT64>
__device__ inline void rich_store(void* base, size_t idx, double v) {
  if constexpr (OUT64) static_cast<double*>(base)[idx] = v;
  else static_cast<float*>(base)[idx] = (float)v;
}

// State vector of one env, any section layout (test_env.py:527-583 order).
//   act(u) -> int, chv(u, i) -> double, hist(u, bin) -> double (count / n or 0),
//   rew(u) -> double, npx(u) -> double (post-move x), py(u) -> double, vel(u) -> double
// Element-wise, 4 or 8 bytes per lane, consecutive lanes on consecutive elements.
template <bool OUT64, typename FAct, typename FChv, typename FHist, typename FRew, typename FNpx, typename FPy,
          typename FVel>
__device__ inline void rich_write_state(const RichParams& r, uint32_t flags, int N, int A, int K, double L, void* out,
                                        size_t row0, int tid, int nthreads, FAct act, FChv chv, FHist hist, FRew rew,
                                        FNpx npx, FPy py, FVel vel) {
  const int S = r.S;
  const int SW = S - r.len_skip;                 // columns this launch writes per row
  if (SW <= 0) return;                           // (a state of sorted distances only)
  const int total = N * SW;
  const int act_w = (flags & DIRAL_F_ACTION_REAL) ? 1 : A;
  int e = tid;
  int u = e / SW, sw = e - u * SW;
  const int du = nthreads / SW, ds = nthreads - du * SW;
  while (e < total) {
    const int s = sw < r.off_skip || r.len_skip == 0 ? sw : sw + r.len_skip;
    double val = 0.0;
    if (r.off_act >= 0 && s >= r.off_act && s < r.off_act + act_w) {
      val = (flags & DIRAL_F_ACTION_REAL) ? (double)act(u) : ((act(u) == s - r.off_act) ? 1.0 : 0.0);   // test_env.py:585-595
    } else if (r.off_chobs >= 0 && s >= r.off_chobs && s < r.off_chobs + A) {
      val = chv(u, s - r.off_chobs);
    } else if (r.off_hist >= 0 && s >= r.off_hist && s < r.off_hist + K) {
      val = hist(u, s - r.off_hist);                                                                  // network.py:501
    } else if (s == r.off_rew) {
      val = rew(u);
    } else if (s == r.off_idx) {
      val = (double)(u + 1);
    } else if (r.off_pos >= 0 && s == r.off_pos) {
      val = npx(u) / L;                                                                               // network.py:403-407
    } else if (r.off_pos >= 0 && s == r.off_pos + 1) {
      val = py(u) / r.H;
    } else if (s == r.off_vel) {
      val = vel(u);
    } else if (r.off_fp >= 0 && s == r.off_fp) {
      val = r.episode;
    } else if (r.off_fp >= 0 && s == r.off_fp + 1) {
      val = r.eps;
    }
    rich_store<OUT64>(out, (row0 + u) * S + s, val);
    e += nthreads; u += du; sw += ds;
    if (sw >= SW) { sw -= SW; u += 1; }
  }
}

// `obs[user][i]` for every (user, resource) of one env into chobs_out, 16 bytes per lane
// (4 floats / 2 doubles) when A allows, consecutive lanes on consecutive quads of a row.
template <bool OUT64, typename FChv>
__device__ inline void rich_write_chobs(void* out, size_t row0, int N, int A, int tid, int nthreads, FChv chv) {
  constexpr int V = OUT64 ? 2 : 4;
  if ((A % V) == 0) {
    const int qpr = A / V, total = N * qpr;
    for (int q = tid; q < total; q += nthreads) {
      const int u = q / qpr, i0 = (q - u * qpr) * V;
      if constexpr (OUT64) {
        double2 v = make_double2(chv(u, i0), chv(u, i0 + 1));
        reinterpret_cast<double2*>(static_cast<double*>(out) + row0 * A)[q] = v;
      } else {
        float4 v = make_float4((float)chv(u, i0), (float)chv(u, i0 + 1), (float)chv(u, i0 + 2), (float)chv(u, i0 + 3));
        reinterpret_cast<float4*>(static_cast<float*>(out) + row0 * A)[q] = v;
      }
    }
  } else {
    for (int e = tid; e < N * A; e += nthreads) {
      const int u = e / A, i = e - u * A;
      rich_store<OUT64>(out, row0 * A + e, chv(u, i));
    }
  }
}

}  // namespace diral
