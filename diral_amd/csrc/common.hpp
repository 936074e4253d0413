// common.hpp - shared host/device definitions for libdiral_env.so (gfx950 only).
//
// Data layout in HBM (one DiralEnv handle, B envs, N vehicles):
//   pos_x,pos_y,vel  f64 [B][N]          Vehicle.pos_x/pos_y/velocity (vehicle.py:9-14)
//   tkey             u32 [B][NR][NV]      neighbour table, SUBJECT-major: tkey[b][k][u] is
//                                        viewer u's entry about vehicle k, packed
//                                        (seq_number << 8) | min(last_updated, 255)
//   tx               f64 [B][NR][NV]      the entry's xpos, same indexing
//   (ypos is not stored: it is pos_y[k] once seq > 0, else 0 - SURVEY.md Q7)
//   la               i32 [B][N][N]       last_arrival_time[tx][rx] (optional)
//   pf               i32 [B][N]          pf_counter (optional)
//   metrics          f64 [B][DIRAL_M_COLUMNS]
// NV = 64 for N <= 64 (else N rounded up to 16); NR = N rounded up to 16, so every wave
// owns 16 existing subject rows and table loads/stores need no predicate.
// The reference stores the table viewer-major (one dict per Vehicle); subject-
// major makes "all viewers of one subject" contiguous, which is what a
// wavefront (lane = viewer) loads with one coalesced instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diral_env.h"

// rare paths (IEEE sqrt, fmod, exp, reward weights) are kept out of line: the fused kernels were
// instruction-fetch bound when these were inlined (profiles/README.md)
#ifndef DIRAL_OUTLINE
#define DIRAL_OUTLINE __attribute__((noinline))
#endif

namespace diral {

constexpr int kModeObserve = 3;  // internal: obtain_state only

constexpr uint32_t kErrAction = 1u;
constexpr uint32_t kErrSeq = 2u;
constexpr uint32_t kErrTable = 4u;   // import: conflicting xpos for one (subject, sequence number)

struct StepParams {
  // geometry
  int B, N, A, K, S, NV, NR;   // NV: padded viewer stride, NR: padded subject rows per env
  uint32_t flags;
  int mode;            // DiralStepMode or kModeObserve
  int reward_design, state_type, posdist_type;
  int age_limit, pf_threshold;
  double pf_penalty;
  double L, H, Rc, Rb, hist_inv_width;   // K / (Rb - (-Rb)): bin-index estimate only
  long long t;
  const long long* t_dev;   // slot clock (diral_env_set_clock) or null: slot number = t + *t_dev
  double episode, eps;
  int out_f64;
  int episode_interval;
  // state-vector section offsets (test_env.py:527-583 order), -1 = absent
  int off_act, off_chobs, off_posdist, off_hist, off_rew, off_idx, off_pos, off_vel, off_fp;
  // persistent state
  double* pos_x;
  double* pos_y;
  double* vel;
  uint32_t* tkey;
  double* tx;
  int32_t* la;
  int32_t* pf;
  double* metrics;
  uint32_t* err;
  const double* edges;  // [K+1] np.linspace(-Rb, Rb, K+1), built on the host
  const double* trace;  // replayed x positions [T][N] or [B][T][N] (network.py:171-178), or null
  int trace_len, trace_per_env;
  // per-call I/O
  const int32_t* actions;
  void* state_out;
  void* rew_out;
  uint8_t* done_out;
  void* chobs_out;
  const double* chobs_in;
  const double* rew_in;
  unsigned long long* dbg;   // phase timestamps (DIRAL_TIMING builds only)
};

// per-handle scratch of the large path (step_large.hpp; diral_env.hip allocates it)
struct LargeScratch {
  unsigned short* src;   // [B][A][N] gather source of (resource, viewer): the closest in-range transmitter, or the viewer itself
  uint32_t* cnt;         // [B][A] transmitters per resource
  unsigned short* alist; // [B][A] the resources with at least one transmitter, ascending
  uint32_t* nact;        // [B] how many
  unsigned char* qflag;  // [B][ceil(N / 2)] per column PAIR: != 0 = it holds (held, as a hint for the next slot) an entry the
                         // thermometer codes do not reach - the rank-keyed merge takes it (large_mergec_kernel /
                         // large_mergen_kernel<CH, 2> write and read)
  double* px0;           // [B][N] positions the slot started with (the own stamp of periodic_update, vehicle.py:63)
  double* rew;           // [B][N] reward per vehicle
  double* rtx;           // [B][N] reception ratio per colliding transmitter (test_env.py:402-405)
};

// LDS carve of the fused step kernel; byte offsets, doubles first.
struct LdsLayout {
  uint32_t px, py, npx, vel, rv, rtx, rew, edges, red, mask, act, inr, hist, cnt, mtab, scratch, total;
};

__host__ __device__ inline uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

__host__ __device__ inline LdsLayout lds_layout(int npad, int A, int K, int vpl, int waves) {
  LdsLayout l;
  uint32_t o = 0;
  l.px = o;    o += 8u * npad;
  l.py = o;    o += 8u * npad;
  l.npx = o;   o += 8u * npad;
  l.vel = o;   o += 8u * npad;
  l.rv = o;    o += 8u * A;
  l.rtx = o;   o += 8u * npad;
  l.rew = o;   o += 8u * npad;
  l.edges = o; o += 8u * (K + 1);
  l.red = o;   o += 8u * 8 * waves;
  l.mask = o;  o += 8u * A * vpl;
  l.act = o;   o += 4u * npad;
  l.inr = o;   o += 4u * npad;
  l.hist = o;  o += 4u * (K | 1) * npad;   // [viewer][K|1]: odd row stride
  l.cnt = o;   o += 4u * npad;
  l.mtab = o;  o += align_up((uint32_t)A * npad, 16);
  l.scratch = o;
  if (vpl > 1) o += 4096u * waves;  // CC columns x npad keys per wave
  l.total = align_up(o, 16);
  return l;
}

}  // namespace diral
