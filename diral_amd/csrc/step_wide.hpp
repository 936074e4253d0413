// step_wide.hpp - the fused env-step kernel specialised for 64 < N <= 256
// vehicles (BASELINE.json configs[2] and [4]: 256-UE/64-res congested,
// 128-UE/64-res dynamic density), the toy YAML's State flags (one-hot action +
// type-2 piggybacked positional histogram), my_step + obtain_state, all y == 0.
//
// Same semantics as step_kernel.hpp (the general path; tests compare the two
// and the oracle bit for bit).  What the profile of the general kernel showed at
// N = 256 (profiles/r01/general_c3_before_wide_summary.txt): one 1024-thread workgroup per CU (115 KB LDS,
// 128 VGPRs + 40 spilled), the gossip merge moving 16-bit (rank, source) keys
// through LDS (4.3 KB per resource step per wave), and a finalize phase of
// ~160 VALU instructions per table entry full of exec-mask control flow.  Here:
//   * 8-bit merge keys.  For a fixed subject k an entry is determined by its
//     sequence number, so the merge only has to carry rank = 255 - lag
//     (lag = t_k - seq against the subject's own fresh sequence number): FOUR
//     columns per 32-bit LDS word, byte-wise max with SDWA.  Exact while every
//     entry of the pass has lag < 255 or seq == 0; otherwise the pass takes a
//     32-bit (seq, source) path, column by column.
//   * no source tracking: after the merge an updated entry needs the xpos that
//     belongs to its final sequence number.  Entries with equal (k, seq) hold
//     equal xpos, so every viewer drops its old xpos into a
//     256-entry LDS table indexed by its OLD rank and updated entries pick
//     theirs up by their NEW rank - one LDS write + one LDS read per entry.
//   * the per-entry state kept across the merge is 16 bits (old rank, age);
//     8 waves per workgroup (16 columns each at N <= 128, 32 at N <= 256), at most
//     84 VGPRs, three workgroups per CU (52 KB LDS each at N = 256).
//   * branch-free finalize (signed distance x1 - x2 is the histogram value when
//     all y are 0; bin = estimate + edge correction).
#pragma once
#include "common.hpp"
#include "rich_out.hpp"
#include "step_fast64.hpp"
#include "step_kernel.hpp"

namespace diral {

constexpr int kWideMaxA = 64;

struct WideLds {
  uint32_t px, npx, rv, edges, red, mask, act, cnt, hist, mtab, scratch, pbytes, lut, slow, total;
};
#ifndef DIRAL_WIDE_WAVES4
#define DIRAL_WIDE_WAVES4 8              // waves per workgroup at N <= 256 (each owns 256 / waves subject columns)
#endif
// waves per workgroup: 8 x 16 columns (N <= 128), 8 x 32 columns (N <= 256)
__host__ __device__ constexpr int wide_waves(int vpl) { return vpl == 2 ? 8 : DIRAL_WIDE_WAVES4; }
#ifndef DIRAL_WIDE_PC2
#define DIRAL_WIDE_PC2 8                 // subject columns per merge pass, N <= 128
#endif
#ifndef DIRAL_WIDE_PC4
#define DIRAL_WIDE_PC4 8                 // subject columns per merge pass, N <= 256 (4 until the xpos ring freed the registers: C3 -3.5 %)
#endif
// The B operand's table: 256 entries of 8 x bf16 - one 16-byte read per K step; the rows a quarter wave reads are random, half
// the LDS cycles of the product are bank conflicts.  DIRAL_WIDE_NIBBLE_LUT=1: 16 entries of 4 x bf16 instead - 128 bytes, every
// entry in banks of its own, two conflict-free 8-byte reads per K step: bit-exact, and measured SLOWER (C3 1.33 -> 1.36 ms, C5
// +- 0: twice the LDS instructions and two more VALU instructions per K step cost more than the conflicts).
#ifndef DIRAL_WIDE_NIBBLE_LUT
#define DIRAL_WIDE_NIBBLE_LUT 0
#endif
constexpr uint32_t kWideLutBytes = DIRAL_WIDE_NIBBLE_LUT ? 128u : 4096u;
// merge scratch per wave: a pass's rank words (one byte per column and viewer), then the
// rank -> xpos table (256 doubles)
// (N <= 256: + 64 bytes in front of the lag -> xpos table of the packed form's finalize phase, whose lookup of a
// never-heard entry - lag "-1" - lands 8 bytes below its column's row: csrc/step_wide_closure.inc)
__host__ __device__ constexpr uint32_t wide_scratch(int vpl) {
  const uint32_t words = 64u * vpl * (vpl == 2 ? DIRAL_WIDE_PC2 : DIRAL_WIDE_PC4);
  return (words > 2048u ? words : 2048u) + (vpl == 4 ? 64u : 0u);
}
// histogram row stride in 32-bit words: two 16-bit bins per word (counts <= 255), odd stride
__host__ __device__ constexpr int wide_hist_stride(int K) { return ((K + 1) / 2) | 1; }
// row stride of the gather-source table in elements (u32 of 4 source bytes at N <= 256, u16 of
// 2 at N <= 128): 64 lanes + one 32-bit word of padding, so that the RICH output phase
// (lane -> (viewer, resource quad)) spreads over the banks; the merge (lane-contiguous) is
// conflict-free at any stride
__host__ __device__ constexpr int wide_mtab_stride(int vpl) { return vpl == 4 ? 65 : 66; }

__host__ __device__ inline WideLds wide_lds_layout(int vpl, int A, int K, bool packed = false) {
  const uint32_t npad = 64u * vpl;
  WideLds l;
  // The per-wave merge scratch sits at LDS offset 0: wave W's words start at the COMPILE-TIME
  // address 2048 W, which the merge loop (one copy per wave index) folds into the immediate
  // offset of its gathers instead of adding a base register to every gather address.
  l.scratch = 0;
  uint32_t o = wide_scratch(vpl) * wide_waves(vpl);        // 2 KB per wave (4 KB at 16 columns per pass)
  // the packed form's merge (step_wide_closure.inc), at compile-time addresses as well (DS immediate offsets): the
  // reachability matrix P of the slot as bytes [viewer][lane group][K step] (8 KB at N = 256, 2 KB at N = 128) and the
  // 256-entry bits -> 8 x bf16 table of the product's B operand
  l.pbytes = l.lut = o;
  // (+ the closure's own rows of P - npad / 16 bytes x 2 waves per viewer - and one row-ready flag per resource: it runs beside P1)
  if (packed) { l.pbytes = o; o += 8u * vpl * npad; l.lut = o; o += kWideLutBytes; o += 8u * vpl * npad; o += 4u * (uint32_t)kWideMaxA; }
  l.px = o;    o += 8u * npad;
  l.npx = o;   o += 8u * npad;
  l.rv = o;    o += 8u * A;
  l.edges = o; o += 8u * (K + 2);
  l.red = o;   o += 8u * 4 * vpl;
  l.slow = o;  o += 8u;                                  // the env holds a pass beyond the codes: a place among the first blocks of the next launch
  l.mask = o;  o += 8u * A * vpl;
  l.act = o;   o += 4u * npad;
  l.cnt = o;   o += 4u * npad;
  l.hist = o;  o += 4u * wide_hist_stride(K) * npad;   // [viewer][stride]: two bins per word
  l.mtab = o;  o += (uint32_t)A * wide_mtab_stride(vpl) * vpl;   // [resource][lane][slot]: gather source viewer (bytes)
  l.total = align_up(o, 16);
  return l;
}

// byte-wise unsigned max of four packed words (16 ranks) with SDWA.  The four
// words are interleaved so that no instruction consumes the partial result of
// the one right before it; the trailing s_nop covers the first consumer the
// compiler places after the block (it cannot see inside).
__device__ inline void max_u8x16(unsigned int (&a)[4], const unsigned int (&b)[4]) {
#define DIRAL_SDWA_MAX(B)                                                                                          \
  "v_max_u32_sdwa %0, %0, %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %1, %1, %5 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %2, %2, %6 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %3, %3, %7 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t"
  asm(DIRAL_SDWA_MAX(0) DIRAL_SDWA_MAX(1) DIRAL_SDWA_MAX(2) DIRAL_SDWA_MAX(3) "s_nop 0"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
#undef DIRAL_SDWA_MAX
}

// ... and of eight packed words (32 ranks)
__device__ inline void max_u8x32(unsigned int (&a)[8], const unsigned int (&b)[8]) {
#define DIRAL_SDWA_MAX8(B)                                                                                          \
  "v_max_u32_sdwa %0, %0, %8 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %1, %1, %9 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %2, %2, %10 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %3, %3, %11 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %4, %4, %12 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %5, %5, %13 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %6, %6, %14 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
  "v_max_u32_sdwa %7, %7, %15 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t"
  asm(DIRAL_SDWA_MAX8(0) DIRAL_SDWA_MAX8(1) DIRAL_SDWA_MAX8(2) DIRAL_SDWA_MAX8(3) "s_nop 0"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
#undef DIRAL_SDWA_MAX8
}
template <int NK>
__device__ inline void max_u8_words(unsigned int (&a)[NK], const unsigned int (&b)[NK]) {
  if constexpr (NK == 4) max_u8x16(a, b);
  else if constexpr (NK == 8) max_u8x32(a, b);
  else {
    static_assert(NK % 8 == 0, "4, 8 or a multiple of 8 words");
#pragma unroll
    for (int q = 0; q < NK; q += 8)
      max_u8x32(*reinterpret_cast<unsigned int (*)[8]>(&a[q]), *reinterpret_cast<const unsigned int (*)[8]>(&b[q]));
  }
}

// Thermometer codes.  In steady state the lag of an entry behind its subject's
// own sequence number is tiny (C3: <= 4, C2: <= 5, C5: <= 7 for 98 % of the entries -
// profiles/lag_distribution.py), so a pass first tries the 8-level code c(lag) = (0xff << lag) & 0xff
// (0: never heard): the codes form a chain under bit inclusion, the code of the smaller lag is the
// bitwise OR, and ONE v_or_b32 merges the four columns of a word where the byte ranks take four
// SDWA maxes.  Exact while every entry of the pass has lag <= 7 or was never heard; otherwise the
// pass is redone with byte ranks (lag < 255), then with 32-bit keys.
#ifndef DIRAL_WIDE_INFLIGHT
#define DIRAL_WIDE_INFLIGHT 16          // table words a lane has in flight while a pass loads its columns
#endif
// (thermo_codes(): step_fast64.hpp)

// xpos ring (see step_fast64.hpp / aux_kernels.hpp): an entry's xpos is a function of
// (subject, sequence number), the ring keeps every subject's 8 latest stamps, so the finalize phase fills
// the rank -> xpos table of a column from the subject's ring row (8 lanes) instead of having all viewers
// scatter their old xpos into it, and EVERY entry that lags at most 7 reads its xpos there.  The per-entry
// xpos plane - two thirds of the table bytes - is only read for older entries and only written when an
// entry reaches lag 7 (or copies an older one).
// run f(std::integral_constant<int, wave>) - a copy of f per wave index (wave-uniform switch): the wave's
// scratch base becomes an immediate offset of the LDS instructions; f(-1): base in a register
#define DIRAL_WIDE_DISPATCH_WAVE(f)                                      \
  do {                                                                   \
    if (!lds_base_is_zero) { f(std::integral_constant<int, -1>{}); break; } \
    switch (wave) {                                                      \
      case 0: f(std::integral_constant<int, 0>{}); break;                \
      case 1: f(std::integral_constant<int, 1>{}); break;                \
      case 2: f(std::integral_constant<int, 2>{}); break;                \
      case 3: f(std::integral_constant<int, 3>{}); break;                \
      case 4: f(std::integral_constant<int, 4>{}); break;                \
      case 5: f(std::integral_constant<int, 5>{}); break;                \
      case 6: f(std::integral_constant<int, 6>{}); break;                \
      default: f(std::integral_constant<int, 7>{}); break;               \
    }                                                                    \
  } while (0)

// An LDS object by its absolute byte address (register + compile-time constant: the constant goes
// into the DS instruction's immediate offset; going through the `extern __shared__` symbol instead
// leaves a relocated `+ 0` add in front of every access)
template <typename T>
__device__ inline const __attribute__((address_space(3))) T* lds_at(unsigned int byte_addr) {
  return (const __attribute__((address_space(3))) T*)(size_t)byte_addr;
}

// LDS byte address of a __shared__ object (what M0-relative DS instructions take)
__device__ inline unsigned int lds_addr(const void* p) {
  return (unsigned int)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// byte j of the packed gather-source word, shifted left by SH (the LDS byte offset of that
// viewer's rank words), one SDWA shift each instead of extract + shift
template <int VPL, unsigned int SH>
__device__ inline void unpack_src(unsigned int mw, unsigned int (&a)[VPL]) {
  if constexpr (VPL == 4) {
    asm("v_lshlrev_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "s"(SH), "v"(mw));
  } else {
    asm("v_lshlrev_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
        : "=&v"(a[0]), "=&v"(a[1]) : "s"(SH), "v"(mw));
  }
}

// byte BYTE of a word of table indices, times 2^SH: the LDS byte offset of that row of the bits -> bf16 table
// (step_wide_closure.inc), shift and byte extraction in one SDWA instruction
template <int BYTE, unsigned int SH = 4u>
__device__ inline unsigned int lut_row(unsigned int w) {
  unsigned int r;
  static_assert(BYTE >= 0 && BYTE < 4, "byte select");
  if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(SH), "v"(w));
  if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(SH), "v"(w));
  if constexpr (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(SH), "v"(w));
  if constexpr (BYTE == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(SH), "v"(w));
  return r;
}

// Reward of a colliding resource (test_env.py:163-199) for N > 64, positions in
// LDS, all y == 0.  Out of line: runs ~once per colliding resource.
template <int VPL>
__device__ DIRAL_OUTLINE double wide_collision_reward(int rd, uint32_t flags, double L, double Rc, int N,
                                                                  const unsigned long long* mkp, int c,
                                                                  const double* s_px) {
  int wgt = 0;
  if (rd == 1 || ((rd == 2 || rd == 5) && c == 2)) {
    double s = 0.0;                    // calculate_avg_distance (network.py:307-316): combinations order
    int cnt = 0;
    for (int ja = 0; ja < VPL; ++ja) {
      unsigned long long ma = mkp[ja];
      while (ma) {
        const int a = ja * 64 + __builtin_ctzll(ma);
        ma &= ma - 1;
        for (int jb = ja; jb < VPL; ++jb) {
          unsigned long long mb = (jb == ja) ? ma : mkp[jb];
          while (mb) {
            const int b = jb * 64 + __builtin_ctzll(mb);
            mb &= mb - 1;
            s = s + dist2d_leaf(s_px[a], 0.0, s_px[b], 0.0);
            ++cnt;
          }
        }
      }
    }
    const double m = s / (double)cnt;
    if (flags & DIRAL_F_TOY_WEIGHTS) {
      double x_min = L + 1, x_max = -L - 1;      // calculate_norm (network.py:225-246)
      int umin = 0, umax = 0;
      for (int u = 0; u < N; ++u) {
        const double x = s_px[u];
        if (x < x_min) { x_min = x; umin = u; }
        if (x > x_max) { x_max = x; umax = u; }
      }
      wgt = (m == dist2d_leaf(s_px[umin], 0.0, s_px[umax], 0.0));
    } else {
      wgt = (m > Rc);
    }
  }
  if (rd == 1) { const double R = (double)wgt / (double)c; return -1.0 * (1.0 - R); }
  if (rd == 2) return (c == 2) ? 2.0 * (double)wgt - (double)c : 0.0 - (double)c;
  if (rd == 3) { const double R = 1.0 / (double)c; return -1.0 * exp(1.0 - R); }
  if (rd == 4) return 1.0 / (double)c;
  return (c == 2 && wgt == 1) ? 0.0 : -1.0;
}

// The far-entry guard of a flagged pass of the packed form at N <= 128 (step_wide_kernel, `cl_far_guard`: what it decides and
// why that is exact).  `rows`: the closure walk's rows of P in LDS, [half][viewer] 8 bytes - sources 0 .. 63 / 64 .. 127 of
// each viewer; `tcq`: the code words of the wave's four quads [quad][NV]; `tkw`: the env's `tkey` rows [subject][NV];
// `flagged`: bit pch = pass pch (8 columns) is flagged.  Returns the passes whose far entries cannot move this slot.
template <bool FULL>
__device__ __attribute__((noinline)) unsigned int wide_far_guard(const unsigned char* rows, int npad, const unsigned int* tcq,
                                                                 const unsigned int* tkw, int kbase, int N, int NV,
                                                                 unsigned int flagged, int lane) {
  constexpr int VPL = 2, PC = 8;
  typedef __attribute__((ext_vector_type(2))) unsigned int g_u32x2;
  const unsigned int ul = (unsigned int)lane;
  g_u32x2 plo[VPL], phi[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    plo[j] = *reinterpret_cast<const g_u32x2*>(rows + 8u * (ul + 64u * j));
    phi[j] = *reinterpret_cast<const g_u32x2*>(rows + 8u * (unsigned int)npad + 8u * (ul + 64u * j));
  }
  unsigned int stable = 0u;
#pragma unroll 1
  for (int pch = 0; pch < 2; ++pch) {
    if (((flagged >> pch) & 1u) == 0u) continue;
    bool viol = false;
    unsigned int qfar = 0u;                     // bit w: quad w of the pass holds an entry beyond the codes (a sequence number in `tkey`)
    unsigned int cw[2][VPL];                   // the pass's raw code words (two quads x VPL slots)
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int j = 0; j < VPL; ++j) cw[w][j] = tcq[(unsigned int)((2 * pch + w) * NV) + ul + 64u * j];
    // the sequence numbers of the pass's far entries, all 16 loads in flight together (a column at a time the guard paid a
    // round trip to HBM per column: 110 k cycles for a workgroup with two flagged passes)
    unsigned int seqa[PC][VPL];
    bool isfa[PC][VPL];
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      const int k = kbase + pch * PC + c;
      const unsigned int* const tkrow = tkw + (size_t)(FULL || k < N ? k : kbase) * NV;
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const unsigned int u = ul + 64u * j;
        const unsigned int rc = (cw[c >> 2][j] >> (8 * (c & 3))) & 255u;
        // beyond the codes once this slot's stamp is taken: raw code 0 (8 or more behind, or never heard) or 0x80 (7 behind:
        // handed over to `tkey` when it got there, vehicle.py:56-70 makes it 8 now)
        isfa[c][j] = (rc & 0x7fu) == 0u && (FULL || (u < (unsigned int)N && k < N));
        seqa[c][j] = tkrow[u] >> 8;              // (unconditional: a padded viewer reads inside the allocation, its value is masked)
      }
    }
#pragma unroll
    for (int c = 0; c < PC; ++c)
#pragma unroll
      for (int j = 0; j < VPL; ++j) seqa[c][j] = isfa[c][j] ? seqa[c][j] : 0u;
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      unsigned int seqv[VPL];
      unsigned long long farm[VPL];
      bool isf[VPL];
#pragma unroll
      for (int j = 0; j < VPL; ++j) { isf[j] = isfa[c][j]; seqv[j] = seqa[c][j]; farm[j] = __ballot(isf[j]); }
      if ((farm[0] | farm[1]) == 0ull) continue;
      if ((__ballot(isf[0] && seqv[0] != 0u) | __ballot(isf[1] && seqv[1] != 0u)) != 0ull) qfar |= 1u << (c >> 2);
      unsigned long long rem[VPL];
#pragma unroll
      for (int j = 0; j < VPL; ++j) rem[j] = farm[j];
      while ((rem[0] | rem[1]) != 0ull) {          // over the distinct far numbers of the column (one, as a rule: nothing to propagate)
        const int js = rem[0] != 0ull ? 0 : 1;
        const unsigned int d = (unsigned int)__builtin_amdgcn_readlane((int)(js == 0 ? seqv[0] : seqv[1]), __builtin_ctzll(rem[js]));
        unsigned long long g[VPL];
        bool lower[VPL];
        bool anylower = false;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          g[j] = __ballot(isf[j] && seqv[j] == d);
          rem[j] &= ~g[j];
          lower[j] = isf[j] && seqv[j] < d;
          anylower = anylower || lower[j];
        }
        if (__ballot(anylower) == 0ull) continue;  // (uniform) nobody holds an older number than d
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const unsigned int hit = (plo[j][0] & (unsigned int)g[0]) | (plo[j][1] & (unsigned int)(g[0] >> 32)) |
                                   (phi[j][0] & (unsigned int)g[1]) | (phi[j][1] & (unsigned int)(g[1] >> 32));
          viol = viol || (lower[j] && hit != 0u);
        }
      }
    }
    // a stable pass: bit pch; its quads without a far entry any more (refreshed since they were flagged): bits 8 + quad -
    // the caller takes their flags down (the coded path only ever raises them: at a hand-over)
    if (__ballot(viol) == 0ull) stable |= (1u << pch) | ((~qfar & 3u) << (8 + 2 * pch));
  }
  return stable;
}

// (uniform_ptr / global_ptr: step_fast64.hpp)
#ifdef DIRAL_TIMING
#define DIRAL_WSTAMP(i) do { if (lane == 0 && p.dbg) p.dbg[((size_t)b * WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DIRAL_WCLOCK(v) v = __builtin_amdgcn_s_memtime()
#else
#define DIRAL_WSTAMP(i) do {} while (0)
#define DIRAL_WCLOCK(v) do {} while (0)
#endif
#ifndef DIRAL_WIDE_MINWAVES2
#define DIRAL_WIDE_MINWAVES2 8           // N <= 128: 64 VGPRs, four 512-thread workgroups per CU (35 KB of LDS each).  Before the xpos ring
                                         // freed the xpos prefetch registers: 1.61 / 1.68 / 1.76 ms for 6 / 7 / 8; with it 1.375 / 1.386 / 1.349 ms
                                         // (with the channel observation 1.46 / 1.47 / 1.33)
#endif
#ifndef DIRAL_WIDE_MINWAVES4
#define DIRAL_WIDE_MINWAVES4 6           // N <= 256: 84 VGPRs, three 512-thread workgroups per CU
#endif
#ifndef DIRAL_WIDE_MINWAVES4P
#define DIRAL_WIDE_MINWAVES4P 4          // ... the packed form: 128 VGPRs (the product's A operand alone takes 64), two workgroups per CU
#endif
#ifndef DIRAL_WIDE_INV_LDS
#define DIRAL_WIDE_INV_LDS 1             // the float32 state vector's 1 / n per viewer through LDS (fetched in the count pass) instead of a global load in P4
#endif
#ifndef DIRAL_WIDE_FUSED_FLAGGED
#define DIRAL_WIDE_FUSED_FLAGGED 1       // packed form: a flagged pass reads and writes the code words itself (no round trip through `tkey`)
#endif
#ifndef DIRAL_WIDE_FLAG_UNROLL
#define DIRAL_WIDE_FLAG_UNROLL 1         // column loops of a flagged pass's unpack / repack stages (4 - the four loads of a word in flight together - measured C5 + 4 %: registers)
#endif
#ifndef DIRAL_WIDE_FAR_GUARD
#define DIRAL_WIDE_FAR_GUARD 1           // packed form at N <= 128: a flagged pass whose far entries provably stay put this slot runs on the coded path
#endif
#ifndef DIRAL_WIDE_EARLY_P3
#define DIRAL_WIDE_EARLY_P3 0            // packed form, my_step: 1 = the P1 waves run P3's prologue + the A operand of pass 0 in front of the P1 barrier;
                                         // 2 = the two walking waves too, before their walk.  Measured on C3 (one box, interleaved): 0: 1.247 ms,
                                         // 1: 1.290, 2: 1.242 - the work costs in front of the barrier what it saves behind it (the kernel is paced
                                         // by the instructions it issues, not by who waits where): off
#endif
#ifndef DIRAL_WIDE_P4V2
#define DIRAL_WIDE_P4V2 1                // the output tail: channel observation written by each wave right behind its P3 (branch-free, the LDS reads of a
                                         // piece in flight together), neighbour counts + 1 / n per WAVE for the rows it writes (one barrier less, no table load)
#endif
#ifndef DIRAL_WIDE_FIN_FMA
#define DIRAL_WIDE_FIN_FMA 1             // packed form's finalize: the fixed-point bin as one v_fma_f64 per entry, the histogram word from its bits
#endif
#ifndef DIRAL_WIDE_FIN_UNROLL
#define DIRAL_WIDE_FIN_UNROLL 0          // packed form's finalize: the quad loop of a viewer slot unrolled (static indices instead of rotating the words)
#endif
#ifndef DIRAL_WIDE_MINWAVES2P
#define DIRAL_WIDE_MINWAVES2P 6          // N <= 128, packed form: 84 VGPRs, three workgroups per CU (43 KB of LDS each)
#endif

// FULL: N == 64 * VPL (every viewer slot and subject row exists): the u < N / k < N predicates
// are compiled out (BASELINE.json's 128- and 256-vehicle configurations)
// CH: my_step_ch (PRR reward, test_env.py:351-443) instead of my_step, as in step_fast64.hpp
// EXTRA: run-time switches for my_step_design and the arrival stamps, as in step_fast64.hpp
// RICH: the output tail of rich_out.hpp (channel observation output, cheap State flags)
// PACKED: the table form (the host decides per handle, csrc/diral_env.hip `use_packed_table`): codes + ages + own sequence
// numbers, or the round-2 (seq, age) plane `tkey` whose passes re-derive the lags every slot and fall back to byte
// ranks in place.  Dense topologies (BASELINE configs[2] and [4]) run packed - the coded passes merge as reachability
// closure + one bf16 product, step_wide_closure.inc -; where most entries lag their subject by more than 7 stamps (sparse
// topologies) every pass of the packed form would detour through the planes - those handles keep the plane form.
template <int VPL, bool OUT64, bool FULL, bool CH, bool EXTRA, bool RICH, bool PACKED>
__global__ __launch_bounds__(64 * wide_waves(VPL), VPL == 2 ? (PACKED ? DIRAL_WIDE_MINWAVES2P : DIRAL_WIDE_MINWAVES2) : (PACKED ? DIRAL_WIDE_MINWAVES4P : DIRAL_WIDE_MINWAVES4)) void step_wide_kernel(const FastParams p, const RichParams r) {
  constexpr int NPAD = 64 * VPL, WAVES = wide_waves(VPL), THREADS = 64 * WAVES;
  constexpr int CPW = NPAD / WAVES;            // subject columns per wave
  constexpr int PC = VPL == 2 ? DIRAL_WIDE_PC2 : DIRAL_WIDE_PC4;   // subject columns per pass
  constexpr int NW = PC / 4;                   // packed rank words (4 columns each) per viewer slot
  constexpr int NK = NW * VPL;                 // ... per lane
  // merge words in LDS: one NW-word vector per viewer, gathered with ONE 8-byte read per slot and step (the plane layout
  // [word][viewer] with 4-byte gathers was dropped: C5 +2 %)
  static_assert(PC % 4 == 0 && CPW % PC == 0 && NPAD * NW * 4 <= wide_scratch(VPL) && wide_scratch(VPL) % 16 == 0, "a pass's rank words fill at most the wave's scratch");
  constexpr uint32_t SCR = wide_scratch(VPL);
  static_assert(WAVES >= VPL && WAVES <= 8, "P2 runs on the first VPL waves; the merge loop has 8 per-wave copies");
  static_assert(8 * PC <= 64, "xpos ring: one lane per (column, lag) of a pass");
  constexpr int FIN_UNROLL = VPL == 2 ? 8 : 2;   // finalize column loop: N <= 128 fully unrolled, N <= 256 by two (VGPR budget)
  constexpr int MT = wide_mtab_stride(VPL);    // gather-source table row stride (elements)
  static_assert(VPL == 2 || VPL == 4, "one lane holds 2 or 4 viewers");
  typedef typename std::conditional<VPL == 4, uint32_t, uint16_t>::type mword_t;

  extern __shared__ __align__(16) unsigned char smem[];
  // The first eight words of the argument block (N ... age_limit), each through a scalar load of its OWN: read as `p.N`
  // etc. the compiler fetches them with one s_load_dwordx8 and - once the hot loops have pushed the tuple out of the
  // SGPR file - reloads all eight lanes (v_readlane, a VALU instruction each) at every use of any one of them:
  // 43 x 8 reloads in the N <= 128 kernel, 16 per table column in its finalize loop for the age limit alone.
  auto late_i32 = [](size_t off) -> int {
    return *(const __attribute__((address_space(4))) int*)(late_kernarg_base() + off);
  };
  const int N = late_i32(offsetof(FastParams, N)), A = late_i32(offsetof(FastParams, A)), K = late_i32(offsetof(FastParams, K));
  const int NV = late_i32(offsetof(FastParams, NV)), NRows = late_i32(offsetof(FastParams, NR));
  const int age_limit = late_i32(offsetof(FastParams, age_limit));
  const int reward_design = late_i32(offsetof(FastParams, reward_design));
  const uint32_t pflags = (uint32_t)late_i32(offsetof(FastParams, flags));
  auto late_f64 = [](size_t off) -> double {
    return *(const __attribute__((address_space(4))) double*)(late_kernarg_base() + off);
  };
  const double pL = late_f64(offsetof(FastParams, L)), pRc = late_f64(offsetof(FastParams, Rc)), pRb = late_f64(offsetof(FastParams, Rb));
  const WideLds lay = wide_lds_layout(VPL, A, K, PACKED);
  double* s_px = reinterpret_cast<double*>(smem + lay.px);
  double* s_npx = reinterpret_cast<double*>(smem + lay.npx);
  double* s_rv = reinterpret_cast<double*>(smem + lay.rv);
  double* s_edges = reinterpret_cast<double*>(smem + lay.edges);
  double* s_red = reinterpret_cast<double*>(smem + lay.red);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem + lay.mask);
  int* s_act = reinterpret_cast<int*>(smem + lay.act);
  unsigned int* s_cnt = reinterpret_cast<unsigned int*>(smem + lay.cnt);
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + lay.hist);
  mword_t* s_mtab = reinterpret_cast<mword_t*>(smem + lay.mtab);
  // my_step_ch per-transmitter scratch (reception ratio R, receivers in range): viewer u's
  // values live in the merge scratch of wave u / 64, which is idle until that wave - the
  // one that reads them in P2 - starts its own P3
  auto rtx_of = [&](int u) -> double* {
    return reinterpret_cast<double*>(smem + lay.scratch + SCR * (u >> 6)) + (u & 63);
  };
  auto inr_of = [&](int u) -> int* {
    return reinterpret_cast<int*>(smem + lay.scratch + SCR * (u >> 6) + 512u) + (u & 63);
  };

  // which env: blocks = envs in order, or (slow envs first, FastParams::slow_*: step_fast64.hpp) the listed envs in the
  // first fast_slow_max(B) blocks.  At N <= 128 on a highway of configs[4]'s density one env in ten has broken into
  // clusters that no longer hear each other: nearly all its passes leave the codes (byte ranks through the planes) and
  // its workgroup lives 3-4 times as long as the others' - dispatched in batch order the last of them end the launch late.
  int b = blockIdx.x;
  unsigned int listed = 0u;                  // (ordinary block) != 0: this env ran in one of the first blocks
  if (p.slow_cnt_r) {
    const int smax = fast_slow_max(p.B);
    if (blockIdx.x < (unsigned int)smax) {
      if (blockIdx.x >= *p.slow_cnt_r) return;
      b = (int)p.slow_list_r[blockIdx.x];
    } else {
      b = (int)blockIdx.x - smax;
      listed = p.slow_flag_r[b];             // a scalar load in flight next to the loads of P0; tested before any global store
    }
  }
  unsigned int* const s_slow = reinterpret_cast<unsigned int*>(smem + lay.slow);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KP = wide_hist_stride(K);          // histogram row stride (words, two bins each)
  const size_t bN = (size_t)b * N;
  const size_t bR = (size_t)b * NRows;
  DIRAL_WSTAMP(0);

  // ---- P0: per-vehicle state and the post-move position into LDS --------------
  int x_unsafe = 0;      // a position that is neither 0 nor at least 2^-447 in magnitude (see p1_fast, step_fast64.hpp)
  if (tid < NPAD) {
    const int u = tid;
    const bool lv = u < N;
    const size_t vi = bN + (lv ? u : 0);
    int a = p.actions[vi];
    double x = p.pos_x[vi];
    const double v = p.vel[vi];
    if (!lv) { a = -1; x = 0.0; }
    if (lv && (a < 0 || a >= A)) { atomicOr(p.err, kErrAction); a = -1; }
    s_act[u] = a;
    s_px[u] = x;
    {
      const unsigned int xh = (unsigned int)__double2hiint(x) & 0x7fffffffu;
      x_unsafe = !(xh >= 0x24000000u || (xh | (unsigned int)__double2loint(x)) == 0u);
    }
    double nx = lv ? py_mod_pos(x + v + pL, pL) : 0.0;     // network.py:203
    if (EXTRA && p.trace && lv) {                            // replay branch, network.py:194-199
      long long tt = (p.t + (p.t_dev ? *p.t_dev : 0ll)) % p.trace_len;
      if (tt < 0) tt += p.trace_len;
      const size_t base = p.trace_per_env ? (size_t)b * p.trace_len : 0;
      nx = p.trace[(base + (size_t)tt) * N + u];
    }
    if (EXTRA && p.nomove) nx = x;                           // network.py:302-305: no mobility, no move
    s_npx[u] = nx;
    s_cnt[u] = 0u;
  }
  for (int j = tid; j < KP * NPAD; j += THREADS) s_hist[j] = 0u;
  if (tid <= K + 1) s_edges[tid] = p.edges[tid < K ? tid : K];
  if (tid == THREADS - 1) s_slow[0] = 0u;
  // PACKED: the closure of the slot's gossip runs beside P1 (below): compile-time LDS addresses behind the merge scratch
  // (step_wide_closure.inc), the bits -> 8 x bf16 table of the product and the row-ready flags of the gather table
  constexpr unsigned int kClPb = wide_scratch(VPL) * WAVES, kClLut = kClPb + 8u * VPL * NPAD, kClRows = kClLut + kWideLutBytes,
                         kClFlag = kClRows + 8u * VPL * NPAD;
  constexpr int P1W = PACKED ? WAVES - 2 : WAVES;         // waves that run P1 (PACKED: the last two walk the closure)
  if constexpr (PACKED) {
    if (tid < (DIRAL_WIDE_NIBBLE_LUT ? 16 : 256)) {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      u32x4 t;                                              // entry e, element j = bit j of e, 0.0 / 1.0
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        t[jj] = ((((unsigned int)tid >> (2 * jj)) & 1u) ? 0x3f80u : 0u) | ((((unsigned int)tid >> (2 * jj + 1)) & 1u) ? 0x3f800000u : 0u);
      if constexpr (DIRAL_WIDE_NIBBLE_LUT) reinterpret_cast<u32x2*>(smem + kClLut)[tid] = u32x2{t[0], t[1]};
      else reinterpret_cast<u32x4*>(smem + kClLut)[tid] = t;
    } else if (tid >= 256 && tid < 256 + kWideMaxA) {
      reinterpret_cast<unsigned int*>(smem + kClFlag)[tid - 256] = 0u;
    }
  }
  // (the barrier P1 needs anyway, carrying one bit: every position of the env is 0 or >= 2^-447, so every nonzero
  // |x_w - x_u| is >= 2^-499 and IS the reference's sqrt(fl(dx^2)) - the search runs without the per-pair exponent test)
  // (carried through the `s_red` slots, free until P2: __syncthreads_or would bring static LDS, and the merge loop
  // relies on the dynamic segment starting at LDS address 0)
  if (listed) return;                        // (uniform; nothing has left the workgroup yet)
  if (tid < NPAD) {
    const unsigned long long uns = __ballot(x_unsafe != 0);
    if (lane == 0) reinterpret_cast<int*>(s_red)[wave] = uns != 0ull ? 1 : 0;
  }
  __syncthreads();
  bool p1_fast = true;
#pragma unroll
  for (int w = 0; w < VPL; ++w) p1_fast = p1_fast && reinterpret_cast<const int*>(s_red)[w] == 0;
  DIRAL_WSTAMP(1);

  int myact[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) myact[j] = s_act[lane + 64 * j];

  // ---- (PACKED) the prologue of P3 - step_wide_closure.inc - as two pieces that depend on nothing this slot computes
  // after P0: the wave's columns (which passes are clean, fresh sequence numbers stamped, the [column][lag] -> xpos
  // table from the ring rows, the ring stamped) and the product's A operand of a pass (the raw code words of the pass's
  // 16 subjects over all sources, as bf16 powers of two with the stamp folded in).  DIRAL_WIDE_EARLY_P3: the P1 waves run
  // both in front of the barrier that ends P1, where they otherwise wait for the closure walk; the two walking waves build
  // theirs before the first rows of the gather table are ready.  (my_step without the run-time extras only: my_step_ch /
  // the EXTRA switches park per-transmitter values in the merge scratch until P2.)
  typedef __attribute__((ext_vector_type(4))) unsigned int cl_u32x4;
  constexpr int CL_KS = NPAD / 32;
  cl_u32x4 cl_a[CL_KS];
  unsigned int cl_passbits = 0u, cl_tkov = 0u;
  bool cl_ovf = false, cl_done = false, cl_a_ready = false;
  unsigned int cl_stable = 0u;                             // bit pch: a flagged pass the guard below found stable this slot
  auto cl_prologue = [&]() {
    const int kbase = wave * CPW;
    const unsigned int ul = (unsigned int)lane;
    cl_done = true;
    if (kbase >= NRows) return;                            // (uniform) waves past the last subject row only help with the closure
    const LateFastArgs la = (LateFastArgs)late_kernarg_base();
    const global_ptr<double> ringp = uniform_ptr(la->ring, 0);
    unsigned char* const xt2 = smem + lay.scratch + wide_scratch(VPL) * wave + 64;   // [column][lag] -> xpos, 8 doubles per column
    const size_t qrow0 = (size_t)b * (NRows >> 2) + (kbase >> 2);
    unsigned int passbits = 0u;
    const global_ptr<const unsigned int> tof = uniform_ptr<const unsigned int>(la->told, qrow0);
#pragma unroll
    for (int pch = 0; pch < CPW / PC; ++pch) {
      unsigned int anyold = 0u;
      if (FULL || kbase + pch * PC < NRows) anyold = tof[2 * pch] | tof[2 * pch + 1];
      passbits |= (__builtin_amdgcn_readfirstlane((int)anyold) != 0 ? 1u : 0u) << pch;
    }
    passbits &= ~cl_stable;                                // (a flagged pass whose far entries cannot move this slot runs coded: cl_far_guard)
    const global_ptr<unsigned int> tsrow = uniform_ptr(la->tseq, bR + kbase);
    const bool cv = ul < (unsigned int)CPW && (FULL || kbase + (int)ul < NRows);
    const unsigned int ts = tsrow[cv ? ul : 0u];
    const unsigned int tkov = cv ? ts + 1u : 0u;
    if (cv && ((passbits >> (ul >> 3)) & 1u) == 0u) tsrow[ul] = tkov;      // (flagged passes stamp their own)
    cl_ovf = cv && tkov >= (1u << 24) - 1u;
#pragma unroll
    for (int i = 0; i < CPW / 8; ++i) {
      const unsigned int c = 8u * i + (ul >> 3), lag = ul & 7u;
      const int k = kbase + (int)c;
      const bool kvalid = FULL || k < N;
      const bool krow = FULL || k < NRows;
      const unsigned int tkc = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)tkov);
      const double rg = ringp[(size_t)(bR + (krow ? k : kbase)) * 8 + ((tkc - lag) & 7u)];
      const double pxk = s_px[kvalid ? k : 0];
      // lag 0 is this slot's stamp (vehicle.py:61-63: the pre-move position under the fresh number)
      reinterpret_cast<double*>(xt2)[c * 8u + lag] = (lag == 0u) ? pxk : rg;
      if (lag == 0u && kvalid && ((passbits >> i) & 1u) == 0u) ringp[(size_t)(bR + k) * 8 + (tkc & 7u)] = pxk;
    }
    cl_passbits = passbits;
    cl_tkov = tkov;
  };
  // ---- (PACKED, N <= 128) the far-entry guard of a flagged pass.  A pass is flagged while one of its quads holds an entry
  // beyond the codes (8 or more stamps behind its subject: code 0, the sequence number in `tkey`); the coded merge -
  // closure + product - cannot carry such values, so a flagged pass used to run the 64-step chain on byte ranks
  // (step_wide_pass.inc) at 2.5 x the cost of a coded pass.  But far values almost never MOVE: on a highway that broke into
  // clusters, the viewers of one cluster hold stale entries about the vehicles of another, all of them the last stamp that
  // crossed, and they hear nobody who knows better - of the flagged passes of BASELINE configs[4] 3 in 10 000 see a far
  // value propagate in a slot (profiles/r06/far_propagation.txt: counted on the oracle).  Vehicle.received_update moves
  // a far value into viewer v's entry about k only if some source s whose entries reach v within this slot - bit s of row
  // v of the closure P, which the walk beside P1 has just computed for the whole env - holds a far entry about k with a
  // HIGHER sequence number than v's own far (or never-heard) one.  The guard asks exactly that, per column, over the
  // distinct far numbers of the column (1-2 as a rule: G = the viewers holding number d, as two ballots; a viewer below d
  // with P[v] & G != 0 would receive it); if no column of the pass has such a pair, every entry that is far and stays
  // uncoded keeps its number, xpos and (incremented) age, every other entry is what the coded merge makes it - the pass
  // runs on the coded path, its quads stay flagged.  A viewer that ends the slot with a coded entry makes the test
  // conservative, never wrong; a failed guard costs the pass its old price plus the test.
  auto cl_far_guard = [&]() -> unsigned int {
    if constexpr (PACKED && VPL == 2 && DIRAL_WIDE_FAR_GUARD) {
      const int kbase = wave * CPW;
      if (kbase >= NRows) return 0u;
      const LateFastArgs la = (LateFastArgs)late_kernarg_base();
      const size_t qrow0 = (size_t)b * (NRows >> 2) + (kbase >> 2);
      const global_ptr<const unsigned int> tof = uniform_ptr<const unsigned int>(la->told, qrow0);
      unsigned int flagged = 0u;
#pragma unroll
      for (int pch = 0; pch < CPW / PC; ++pch) {
        unsigned int anyold = 0u;
        if (FULL || kbase + pch * PC < NRows) anyold = tof[2 * pch] | tof[2 * pch + 1];
        flagged |= (__builtin_amdgcn_readfirstlane((int)anyold) != 0 ? 1u : 0u) << pch;
      }
      if (flagged == 0u) return 0u;
      // (out of line: inlined, its registers cost every workgroup - also the nine in ten that never get here - spills in the
      // output tail behind it: + 35 k cycles per workgroup)
      const unsigned int g = wide_far_guard<FULL>(smem + kClRows, NPAD, la->tcode + qrow0 * NV, la->tkey + bR * NV, kbase, N, NV, flagged, lane);
      // (quads of a stable pass that hold no far entry any more: unflagged here - a hand-over in this slot's finalize raises
      // the flag again, behind this store in the wave's own order)
      if ((g >> 8) != 0u && lane < CPW / 4 && (((g >> 8) >> lane) & 1u) != 0u) la->told[qrow0 + lane] = 0u;
      return g & 0xffu;
    }
    return 0u;
  };
  auto cl_build_a = [&](int pass) {
    const int kbase = wave * CPW;
    const int c16 = lane & 15, g4 = lane >> 4;             // product: lane = (subject or viewer of the tile, K group)
    const int NQ = NRows >> 2;
    const int kk = kbase + 16 * pass + c16;                        // this lane's subject (row of the product)
    const int qa = (kk >> 2) < NQ ? (kk >> 2) : NQ - 1;            // (rows past the table: any row, never used)
    const unsigned int sh = 8u * (unsigned int)(kk & 3);
    const global_ptr<const unsigned int> trow0 =
        uniform_ptr<const unsigned int>(((LateFastArgs)late_kernarg_base())->tcode, (size_t)b * NQ * NV);
    const unsigned int off0 = (unsigned int)qa * (unsigned int)NV + 8u * (unsigned int)g4;
    // sources 32 s + 8 g + (0 .. 7): eight consecutive words of the subject's quad row (a row shorter than 256
    // viewers: the words behind it, inside the allocation - their P bits are 0).  Four K steps = eight 16-byte loads
    // in flight at a time
    typedef const __attribute__((address_space(1))) cl_u32x4* gv4;
#pragma unroll
    for (int s0 = 0; s0 < CL_KS; s0 += 4) {
      cl_u32x4 w[8];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        w[2 * s] = *(gv4)(trow0 + off0 + 32u * (s0 + s));
        w[2 * s + 1] = *(gv4)(trow0 + off0 + 32u * (s0 + s) + 4u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // Vehicle.periodic_update folded in: every lag + 1 = the code shifted, its popcount that of (raw & 0x7f)
        auto bf = [&](unsigned int lo, unsigned int hi) -> unsigned int {
          return ((unsigned int)__popc((lo >> sh) & 0x7fu) << 11) | ((unsigned int)__popc((hi >> sh) & 0x7fu) << 27);
        };
        const cl_u32x4 w0 = w[2 * s], w1 = w[2 * s + 1];
        cl_a[s0 + s][0] = bf(w0.x, w0.y); cl_a[s0 + s][1] = bf(w0.z, w0.w); cl_a[s0 + s][2] = bf(w1.x, w1.y); cl_a[s0 + s][3] = bf(w1.z, w1.w);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the own entry of the subject (source == subject): lag 0.  Subject kbase + 16 pass + c: K step (kbase + 16 pass) / 32,
    // lane group ((kbase + 16 pass) / 8) % 4 + (c >> 3), element c & 7 - a switch over the K step (static register indices)
    const int kk0 = kbase + 16 * pass;
    unsigned int om[4];
#pragma unroll
    for (int vi = 0; vi < 4; ++vi)
      om[vi] = (g4 == ((kk0 >> 3) & 3) + (c16 >> 3) && ((c16 & 7) >> 1) == vi) ? (0xffffu << (16 * (c16 & 1))) : 0u;
    auto fix_own = [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
#pragma unroll
      for (int vi = 0; vi < 4; ++vi) cl_a[W][vi] = (cl_a[W][vi] & ~om[vi]) | (0x40004000u & om[vi]);
    };
    if constexpr (CL_KS == 8) {
      switch (kk0 >> 5) {
        case 0: fix_own(std::integral_constant<int, 0>{}); break;
        case 1: fix_own(std::integral_constant<int, 1>{}); break;
        case 2: fix_own(std::integral_constant<int, 2>{}); break;
        case 3: fix_own(std::integral_constant<int, 3>{}); break;
        case 4: fix_own(std::integral_constant<int, 4>{}); break;
        case 5: fix_own(std::integral_constant<int, 5>{}); break;
        case 6: fix_own(std::integral_constant<int, 6>{}); break;
        default: fix_own(std::integral_constant<int, 7>{}); break;
      }
    } else {
      switch (kk0 >> 5) {
        case 0: fix_own(std::integral_constant<int, 0>{}); break;
        case 1: fix_own(std::integral_constant<int, 1>{}); break;
        case 2: fix_own(std::integral_constant<int, 2>{}); break;
        default: fix_own(std::integral_constant<int, 3>{}); break;
      }
    }
  };
  constexpr bool CL_EARLY = PACKED && !CH && !EXTRA && DIRAL_WIDE_EARLY_P3 != 0;
  auto cl_early = [&]() {
    if (lds_addr(smem) != 0u) return;                       // (the scratch carve below assumes what the P3 code checks)
    cl_prologue();
    if (wave * CPW < NRows && (FULL || wave * CPW < NRows)) { cl_build_a(0); cl_a_ready = true; }
  };

  // ---- P1: per owned resource: transmitter set, closest in-range transmitter
  // per viewer (network.py:378-398: ascending id, strict '<'), gather sources,
  // collision reward --------------------------------------------------------------
  {
    double mypx[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) mypx[j] = s_px[lane + 64 * j];
#pragma unroll 1
    for (int i = wave; i < (wave < P1W ? A : 0); i += P1W) {
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
      auto search = [&](auto fast_tag) {
      constexpr bool ABS = decltype(fast_tag)::value;     // |dx| without the per-pair exponent test (p1_fast)
#pragma unroll
      for (int jt = 0; jt < VPL; ++jt) {
        unsigned long long m = mk[jt];
        while (m) {
          const int wl = __builtin_ctzll(m);
          const int w = jt * 64 + wl;
          m &= m - 1;
          const double xw = readlane_f64(mypx[jt], wl);        // (from the registers: an LDS read here is a round trip per transmitter)
          int n_in = 0;
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            double d, keep;                                   // (`best` holds the SIGNED difference on the fast path: its magnitude
            if constexpr (ABS) {                              // comes from the compares' source modifiers, see step_fast64.hpp)
              keep = mypx[j] - xw;
              d = __builtin_fabs(keep);
            } else {
              d = keep = fast_dist<true>(xw, 0.0, mypx[j], 0.0);
            }
            const bool inr = d < pRc;
            const bool bt = inr && (d < __builtin_fabs(best[j]));
            best[j] = bt ? keep : best[j];
            bid[j] = bt ? w : bid[j];
            if (EXTRA && p.la && (FULL || lane + 64 * j < N) && (myact[j] != i) && !inr)
              p.la[(bN + w) * N + lane + 64 * j] = -1;          // find_closest_tx side effect (network.py:394)
            if ((CH || (EXTRA && p.prr)) && c > 1)              // in_range[tx] (test_env.py:395-397)
              n_in += __popcll(__ballot((FULL || lane + 64 * j < N) && (myact[j] != i) && inr));
            if (EXTRA && !CH && p.design && c > 1)              // my_step_design: tx of this resource within 2 Rc
              n_in += __popcll(__ballot((myact[j] == i) && (lane + 64 * j != w) && (d < 2.0 * pRc)));
          }
          if ((CH || (EXTRA && p.prr)) && c > 1 && lane == 0) *inr_of(w) = n_in;
          if (EXTRA && !CH && p.design && c > 1 && lane == 0) *rtx_of(w) = (n_in == 0) ? 1.0 : -(double)(n_in + 1);   // network.py:122-157
        }
      }
      };
      // my_step without the EXTRA switches: nothing needs "in range" per transmitter, and the closest IN-RANGE transmitter
      // (strict '<', first of equals: network.py:378-398) is the closest of all if that one is in range, else none - a plain
      // running minimum (v_min_f64 on the magnitude) and ONE range test per resource and viewer slot: 4 vector instructions
      // per (transmitter, slot) instead of 8 (step_fast64_body.inc, search_min).  Worth nothing while the merge chain paced the
      // kernel (round 4: +- 0 on C3); with the merge on the matrix pipe P1 is a fifth of a workgroup's life.
      auto search_min = [&](auto fast_tag) {
        constexpr bool ABS = decltype(fast_tag)::value;
#pragma unroll
        for (int jt = 0; jt < VPL; ++jt) {
          unsigned long long m = mk[jt];
          while (m) {
            const int wl = __builtin_ctzll(m);
            const int w = jt * 64 + wl;
            m &= m - 1;
            const double xw = readlane_f64(mypx[jt], wl);
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              double d;
              if constexpr (ABS) d = mypx[j] - xw;                 // signed: the magnitude through source modifiers
              else d = fast_dist<true>(xw, 0.0, mypx[j], 0.0);
              const bool bt = __builtin_fabs(d) < best[j];
              bid[j] = bt ? w : bid[j];
              asm("v_min_f64 %0, %0, |%1|" : "+v"(best[j]) : "v"(d));
            }
          }
        }
#pragma unroll
        for (int j = 0; j < VPL; ++j)
          if (!(best[j] < pRc)) bid[j] = -1;                       // network.py:385-386: none in range
      };
      if constexpr (!CH && !EXTRA) {
        if (p1_fast) search_min(std::true_type{});
        else search_min(std::false_type{});
      } else {
        if (p1_fast) search(std::true_type{});
        else search(std::false_type{});
      }
      unsigned int mw = 0u;
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const int u = lane + 64 * j;
        const bool got = (myact[j] != i) && (bid[j] >= 0) && (u < N);
        mw |= (unsigned int)(got ? bid[j] : u) << (8 * j);
        if (EXTRA && CH && p.la && got) p.la[(bN + bid[j]) * N + u] = (int32_t)(p.t + (p.t_dev ? *p.t_dev : 0ll));   // test_env.py:436
      }
      s_mtab[i * MT + lane] = (mword_t)mw;
      if constexpr (PACKED) {
        // the row is ready (a wave's LDS operations execute in order: whoever sees the flag sees the row)
        wave_lds_order();
        // (through an LDS-address-space pointer: the generic one compiled to flat_store + s_waitcnt vmcnt(0) per resource)
        if (lane == 0) *(volatile __attribute__((address_space(3))) unsigned int*)(size_t)(lds_addr(smem) + kClFlag + 4u * (unsigned int)i) = 1u;
      }
      if (CH || (EXTRA && p.prr)) {
        if (c > 1) {
          // received[tx] = #rx whose nearest in-range tx is tx; R = received / in_range (test_env.py:398-405)
          wave_lds_order();
#pragma unroll
          for (int jt = 0; jt < VPL; ++jt) {
            unsigned long long m2 = mk[jt];
            while (m2) {
              const int w = jt * 64 + __builtin_ctzll(m2);
              m2 &= m2 - 1;
              int n_rec = 0;
#pragma unroll
              for (int j = 0; j < VPL; ++j)
                n_rec += __popcll(__ballot((FULL || lane + 64 * j < N) && (myact[j] != i) && bid[j] == w));
              if (lane == 0) {
                const int n_in = *inr_of(w);
                *rtx_of(w) = n_in > 0 ? (double)n_rec / (double)n_in : 1.0;
              }
            }
          }
        }
      }
      if (!CH && c > 1 && !(EXTRA && p.design)) {               // test_env.py:159-199
        double rw;
        if (reward_design == 2 && !(pflags & DIRAL_F_TOY_WEIGHTS)) {
          if (c == 2) {
            // the two transmitters, ascending (network.py:291-295 weight of a pair)
            int ab[2], n = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              unsigned long long m = mk[j];
              while (m) { ab[n < 2 ? n : 1] = j * 64 + __builtin_ctzll(m); ++n; m &= m - 1; }
            }
            const double dab = fast_dist<true>(s_px[ab[0]], 0.0, s_px[ab[1]], 0.0);
            rw = 2.0 * (double)(dab > pRc) - (double)c;       // (0 + d) / 1 == d exactly
          } else {
            rw = 0.0 - (double)c;
          }
        } else {
          rw = wide_collision_reward<VPL>(reward_design, pflags, pL, pRc, N, s_mask + i * VPL, c, s_px);
        }
        if (lane == 0) s_rv[i] = rw;
      }
    }
  }
  if constexpr (CL_EARLY) {
    if (wave < P1W) cl_early();                              // (the P1 waves: in front of the barrier, beside the closure walk)
  }
  if constexpr (PACKED) {
    // ---- the reachability closure of the slot (step_wide_closure.inc: final = P . stamped, P = (I + E_A) ... (I + E_1)),
    //      walked ONCE per env, BESIDE P1: waves 6 and 7 - 128 source bits each, rows of P in LDS, gather the source's 16
    //      bytes and ds_or them into the own row - take the rows of the gather table in resource order as the six P1 waves
    //      finish them (a flag per row; an idle resource's row is the identity), and leave P as bytes
    //      [viewer][lane group][K step of 32 sources] for the products.  Behind P1, on waves 4-7 with the others waiting at
    //      a barrier for it, the walk was 15 k of a workgroup's 140 k cycles; beside it P1 takes six waves 21 k instead of
    //      eight waves 15.5 k and the walk disappears behind it.
    if (wave >= P1W) {
      if (!(__builtin_amdgcn_readfirstlane(lds_addr(smem)) == 0u)) __builtin_trap();   // (compile-time LDS addresses: dynamic segment at 0)
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      // a wave's share of a row of P: NPAD / 2 source bits = 16 bytes (N <= 256) / 8 bytes (N <= 128) per viewer
      typedef typename std::conditional<VPL == 4, u32x4, u32x2>::type rowv_t;
      constexpr unsigned int RB = 4u * VPL;                // bytes of that share; log2: 4 / 3
      constexpr unsigned int RSH = VPL == 4 ? 4u : 3u;
      const int cwv = wave - P1W;
      unsigned char* const rows = smem + kClRows + RB * NPAD * cwv;
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const int u = lane + 64 * j;                       // identity: bit u of the NPAD-bit row, this wave's half
        rowv_t id;
#pragma unroll
        for (int w = 0; w < VPL; ++w) id[w] = (u >> 5) == VPL * cwv + w ? 1u << (u & 31) : 0u;
        reinterpret_cast<rowv_t*>(rows)[u] = id;
      }
      wave_lds_order();
#if DIRAL_WIDE_EARLY_P3 >= 2
      if constexpr (CL_EARLY) cl_early();                     // (the walking waves: before the first rows of the gather table are ready)
#endif
      const volatile unsigned int* const flag = reinterpret_cast<const volatile unsigned int*>(smem + kClFlag);
      // (the flag and the row of step i + 1 are requested in front of step i's gathers - flag first: in-order LDS queue, a
      // set flag vouches for the row read behind it - so that a walk that lags the P1 waves pays no LDS round trip per
      // step for them; polled with a round trip per step the walk took 36 k cycles against P1's 22 k)
      unsigned int mw_n = 0u, fl_n = 0u;
      __builtin_amdgcn_s_setprio(3);                          // (the walk is the critical path of P1; it shares its SIMD with a P1 wave)
#pragma unroll 1
      for (int i = 0; i < A; ++i) {
        unsigned int mw = mw_n;
        if (fl_n == 0u) {                                     // (uniform) not seen ready yet: poll
          while (flag[i] == 0u) __builtin_amdgcn_s_sleep(1);
          mw = (unsigned int)s_mtab[i * MT + lane];
        }
        const int i1 = i + 1 < A ? i + 1 : i;
        fl_n = flag[i1];
        mw_n = (unsigned int)s_mtab[i1 * MT + lane];
        fl_n = (unsigned int)__builtin_amdgcn_readfirstlane((int)fl_n);
        if (i + 1 >= A) fl_n = 0u;
        unsigned int sa[VPL];
        unpack_src<VPL, RSH>(mw, sa);                         // source viewer * RB: the byte offset of its row
        rowv_t g[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) g[j] = *reinterpret_cast<const rowv_t*>(rows + sa[j]);
        wave_lds_order();
        // (a vehicle without a source gathers its own row: a no-op; the transmitters of this resource are nobody's
        // receivers in this step, so their rows are read as the earlier steps left them - in-order LDS queue)
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          unsigned long long* const own = reinterpret_cast<unsigned long long*>(rows + RB * (lane + 64 * j));
          __hip_atomic_fetch_or(own, ((unsigned long long)g[j][1] << 32) | g[j][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if constexpr (VPL == 4)
            __hip_atomic_fetch_or(own + 1, ((unsigned long long)g[j][3] << 32) | g[j][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_lds_order();
      }
      // P as bytes [viewer][lane group g][K step s]: byte g of dword s of the viewer's row = the sources
      // 32 s + 8 g + (0 .. 7); this wave owns the dwords s = VPL cwv ... VPL cwv + VPL - 1: VPL bytes per lane group
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        const unsigned int u = (unsigned int)lane + 64u * j;
        const rowv_t r = reinterpret_cast<const rowv_t*>(rows)[u];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const unsigned int sel = 0x0c0c0000u | ((4u + gg) << 8) | (unsigned int)gg;
          const unsigned int lo16 = __builtin_amdgcn_perm(r[1], r[0], sel);
          if constexpr (VPL == 4) {
            const unsigned int hi16 = __builtin_amdgcn_perm(r[3], r[2], sel);
            *reinterpret_cast<unsigned int*>(smem + kClPb + u * 32u + gg * 8u + 4u * cwv) = lo16 | (hi16 << 16);
          } else {
            *reinterpret_cast<unsigned short*>(smem + kClPb + u * 16u + gg * 4u + 2u * cwv) = (unsigned short)lo16;
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      wave_lds_order();
    }
  }
  DIRAL_WSTAMP(2);
  __syncthreads();
  DIRAL_WSTAMP(3);

  // ---- P2 (first VPL waves): reward per transmitter, metric partials, positions --
  // (the two stores of P2 - reward, position - moved behind P3, so that the table words P3 asks for first do not wait for
  // them: measured, +- 0 at C3 and C5)
  if (tid < NPAD) {
    const unsigned long long late2 = late_kernarg_base();         // late-bound arguments: see step_fast64.hpp
    const int u = tid;
    double rw = 0.0, prr = 0.0;
    int sole = 0, coll = 0;
    const int a = s_act[u];
    if (u < N && a >= 0) {
      int c = 0;
#pragma unroll
      for (int j = 0; j < VPL; ++j) c += __popcll(s_mask[a * VPL + j]);
      if (CH) {
        const double R = (c > 1) ? *rtx_of(u) : 1.0;                          // test_env.py:411-429
        const bool plain = (reward_design == 2);
        rw = plain ? ((c > 1) ? -1.0 * (1.0 - R) : 1.0) : fast_ch_reward(reward_design, c > 1, R);
        coll = c > 1; sole = !(c > 1); prr = R;
      } else if (c > 1) { rw = (EXTRA && p.design) ? *rtx_of(u) : s_rv[a]; coll = 1; } else { rw = 1.0; sole = 1; }      // test_env.py:211-222, 297-301
      if (!CH && EXTRA && p.prr) prr = (c > 1) ? *rtx_of(u) : 1.0;            // DIRAL_F_TRACK_PRR: the metric only
      if constexpr (RICH && !CH) {                                             // proportional fairness, as in step_fast64.hpp
        const LateRichArgs lr = (LateRichArgs)(late2 + kRichArgOffset);
        int32_t* const pf = lr->pf;
        if (pf && !(EXTRA && p.design)) {
          if (c > 1) {
            const int pc = pf[bN + u];
            if (pc > lr->pf_threshold) rw = lr->pf_penalty;
            pf[bN + u] = pc + 1;
          } else {
            pf[bN + u] = 0;
          }
        }
      }
    }
    // (RICH: also for a vehicle whose action was rejected - the output phase reads the
    // reward column of the state back from rew_out, which therefore must be defined)
    const LateFastArgs lp2 = (LateFastArgs)late2;
    void* const rew_out2 = lp2->rew_out;
    if (u < N && (RICH || a >= 0) && rew_out2) {
      if constexpr (OUT64) static_cast<double*>(rew_out2)[bN + u] = rw;
      else static_cast<float*>(rew_out2)[bN + u] = (float)rw;
    }
    if (u < N) lp2->pos_x[bN + u] = s_npx[u];
    // (counts by ballot, the reward sum by DPP moves, the PRR sum on the shuffle tree of the general kernel: step_fast64.hpp)
    const double vr = wave_sum_f64(rw);
    const int vs = __popcll(__ballot(sole != 0)), vc = __popcll(__ballot(coll != 0));
    double vp = prr;
    if (CH || EXTRA) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) vp += __shfl_down(vp, off);
    }
    if (lane == 0) {
      s_red[wave * 4 + 0] = vr; s_red[wave * 4 + 1] = vp; s_red[wave * 4 + 2] = (double)vs; s_red[wave * 4 + 3] = (double)vc;
    }
  }

  // ---- P3: stamp + gossip merge + xpos + histogram over this wave's 16 columns ---
  unsigned int* const sw = reinterpret_cast<unsigned int*>(smem + lay.scratch + SCR * wave);   // merge words
  double* const xt = reinterpret_cast<double*>(sw);                                             // rank -> xpos
  // (the dynamic LDS segment starts at address 0 when the kernel has no static LDS - checked, not assumed)
  const bool lds_base_is_zero = __builtin_amdgcn_readfirstlane(lds_addr(smem)) == 0u;
  const double inv_w = late_f64(offsetof(FastParams, inv_w));
  const LateFastArgs lpp = (LateFastArgs)late_kernarg_base();      // the packed planes: late-bound kernel arguments
  const global_ptr<double> ringp = uniform_ptr(lpp->ring, 0);
  unsigned int* const g_tcode = lpp->tcode;
  unsigned int* const g_tage = lpp->tage;
  unsigned int* const g_tseq = lpp->tseq;
  unsigned int* const g_told = lpp->told;

  // resources with at least one transmitter, as a wave-uniform bit word (A <= 64)
  unsigned long long actw;
  {
    unsigned long long any = 0ull;
    if (lane < A) {
#pragma unroll
      for (int j = 0; j < VPL; ++j) any |= s_mask[lane * VPL + j];
    }
    actw = __ballot(any != 0ull);
  }
  // ... and per viewer slot j the resources with a transmitter IN that slot: a vehicle's merge words are gathered by
  // others only in the step of the resource it transmits on, so the merge loop writes a slot's words back to LDS right
  // before step i only when the slot holds a transmitter of i (a store costs three times a gather: MI355X_MICROARCH LDS)
  unsigned long long txs[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) txs[j] = __ballot(lane < A && s_mask[(lane < A ? lane : 0) * VPL + j] != 0ull);

  // viewer-side tail of one entry: stores, then its histogram contribution
  // (Network.dist_piggy + get_positional_dist_2_piggy, network.py:538-558, 473-513)
  // neighbour count per viewer: in registers where the VGPR budget has room (N <= 128: one
  // barrier and one pass over the histogram less), else the row sum of the histogram
  constexpr bool REGCNT = VPL == 2 && !PACKED;   // (the packed form's coded finalize counts by the row sum: a per-slot count + one LDS atomic there measured C5 + 5 %)
  unsigned int mycnt[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) mycnt[j] = 0u;
  // xmode 0: xpos is stored for the whole 64-viewer slot as soon as one of its entries changed
  // (unchanged lanes rewrite their value): a lane-masked store leaves partially written
  // 32-byte sectors, which HBM turns into read-modify-write - measured 1.4x the traffic.
  // xmode 1 (xpos ring): only the lanes with `upd` set store (an entry at lag 7 or a copy of an older one: rare);
  // xmode 2: every lane stores.  xmode 3: a coded entry of the packed table - both planes only at the hand-over.
  auto emit = [&](int k, bool kvalid, int j, bool upd, unsigned int wn, double xg, global_ptr<unsigned int> tkrow,
                  global_ptr<double> txrow, auto xmode_tag) {
    constexpr int XMODE = decltype(xmode_tag)::value;
    const int u = lane + 64 * j;
    const bool lv = FULL || ((u < N) && kvalid);
    const bool slot_upd = XMODE == 0 ? (__ballot(upd || u == k) != 0ull) : (XMODE == 2 || upd);   // (1, 4: the lanes with `upd`)
    if constexpr (XMODE == 3) {
      // coded entry (packed table): nothing goes to the planes (the hand-over at lag 7 is the caller's; `wn` is the age)
    } else if (lv) {
      if constexpr (XMODE != 4) tkrow[(unsigned int)u] = wn;     // (4: the flagged pass of the packed form writes the far entries' words itself)
      if (slot_upd) txrow[(unsigned int)u] = xg;
    }
    // all y == 0: v = x1 - x2 IS d * sign exactly, d = |v| - unless the square underflows, which only the comparison
    // with a bin edge at exactly 0 can notice: that case is handled in the rare edge branch (see step_fast64.hpp)
    double v = xg - s_npx[u];
    const bool ok = lv && (u != k) && ((int)(wn & 255u) < age_limit) && (__builtin_fabs(v) < pRb);
    if (ok) {
      bool unsafe;
      int bin = hist_bin_estimate(v, pRb, inv_w, K, unsafe);        // (step_kernel.hpp: the edges are read only near an edge)
      if (unsafe) {
        bin = hist_bin_clamp(bin, K);
        if (((unsigned int)__double2hiint(v) & 0x7fffffffu) < 0x20b00000u) {     // |v| below 2^-500 (its square underflows) or 0
          const double d = dist_general(s_npx[u] - xg, 0.0);
          v = (v > 0.0) ? d : -d;
        }
        const double e0 = s_edges[bin], e1 = s_edges[bin + 1];
        bin += (v >= e1 ? 1 : 0) - (v < e0 ? 1 : 0);
      }
      atomicAdd(&s_hist[u * KP + (bin >> 1)], 1u << (16 * (bin & 1)));
      if constexpr (REGCNT) mycnt[j] += 1u;
    }
  };

  // Table loads are unconditional and unclamped (all 16 of a pass in flight together,
  // one lane offset + immediate slot offsets): a padded viewer slot u >= N reads past
  // the row into the next one - or into the 256-element slack behind the last row
  // (diral_env_create) - and is masked.
  const unsigned int ul = (unsigned int)lane;
  // byte c of the packed per-slot words (c wave-uniform, possibly dynamic)
  // (the column loops below are nests word w (static) x column cc within the word)
  auto pick = [&](const unsigned int (&arr)[NK], int j, int w, int cc) -> unsigned int {
    return (arr[w * VPL + j] >> (8 * cc)) & 255u;
  };

  bool ovf = false;
  unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, acc_load = 0, acc_merge = 0, acc_fin = 0, t_p3 = 0;
  DIRAL_WCLOCK(t_p3);
  // PACKED (dense topologies, BASELINE configs[2] and [4]): codes, ages, own sequence numbers (below).
  // Otherwise the (seq, age) plane `tkey` as in round 2: on sparse topologies most entries lag more than 7 stamps; the
  // 8-level codes cannot carry them, and the packed form's detour through the planes for such passes costs more than it
  // saves (a sparse 256-vehicle highway + 60 %).  At configs[4]'s density one env in ten - a highway that broke into
  // clusters - runs nearly all its passes flagged (profiles/flag_fraction.py) and is dispatched first.
  if constexpr (PACKED) {
  // the coded merge + finalize of the clean passes: reachability closure + one bf16 product on the matrix pipe
  // (leaves `passbits`: bit pch = a quad of pass pch was flagged when the slot began -> the loop below)
#include "step_wide_closure.inc"
  // ---- flagged passes (a quad with an entry beyond the codes): through the planes, in a loop of their own so that
  //      the coded pass above carries none of this path's registers
#pragma unroll 1
  for (int pch = 0; pch < CPW / PC; ++pch) {
    const int kbase = wave * CPW + pch * PC;
    if (kbase >= NRows) break;
    if (EXTRA && RICH && p.notab) break;         // no piggybacked tables (test_env.py:138-139, 231-238): nothing to stamp, merge or observe
    const size_t qrow = (size_t)b * (NRows >> 2) + (kbase >> 2);
    const global_ptr<unsigned int> tcrow = uniform_ptr(g_tcode, qrow * NV);
    const global_ptr<unsigned int> tarow = uniform_ptr(g_tage, qrow * NV);
    const global_ptr<unsigned int> tsrow = uniform_ptr(g_tseq, bR + kbase);
    {
      // (the flags as the previous slot left them: the coded loop above only ever SETS flags of clean quads it
      // handed an entry over in - those passes ran there and must not run again)
      if (((passbits >> pch) & 1u) == 0u) continue;
#ifdef DIRAL_WIDE_NO_FLAGGED
      continue;   // (compile-time probe: the coded loop's own register needs)
#endif
    }
    // ---- flagged pass: through the planes ------------------------------------------------------------------
    // (timing builds: unpack / plane pass / repack of the flagged passes go to the load / merge / finalize accumulators)
    unsigned long long tf0 = 0, tf1 = 0, tf2 = 0, tf3 = 0;
    DIRAL_WCLOCK(tf0);
    // the pass's entries as (seq, age) words in `tkey` / the packed words again from `tkey`: the round trip of round 4's
    // flagged pass (unpack 35 k + repack 20 k cycles per wave beside a plane pass of 75 k, profiles/r05/phase_timing_wide.txt).
    // DIRAL_WIDE_FUSED_FLAGGED: the pass builds its lag bytes from the code words and writes code words back itself
    // (step_wide_pass.inc, DIRAL_PASS_PACKED_IO) - `tkey` is read for code-0 entries and written for entries 7 or more
    // behind only - and the two stages below serve the 32-bit path alone (it walks `tkey` column by column).
    auto unpack_pass = [&]() {
      {
        // the pass's entries as (seq, age) words, unstamped: coded ones from the subject's own number and the lag,
        // code-0 ones keep the sequence number `tkey` holds (0: never heard); ages from the age words
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          unsigned int cwj[VPL], awj[VPL];
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            cwj[j] = tcrow[(unsigned int)(w * NV) + ul + 64u * j];
            awj[j] = tarow[(unsigned int)(w * NV) + ul + 64u * j];
          }
#pragma unroll DIRAL_WIDE_FLAG_UNROLL
          for (int cc = 0; cc < 4; ++cc) {
            const int k = kbase + 4 * w + cc;
            if (!(FULL || k < N)) continue;
            const global_ptr<unsigned int> tkrow = uniform_ptr(p.tkey, (bR + k) * NV);
            const unsigned int ts_old = tsrow[4 * w + cc];
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const int u = lane + 64 * j;
              if (FULL || u < N) {
                const unsigned int r = (cwj[j] >> (8 * cc)) & 255u, a = (awj[j] >> (8 * cc)) & 255u;
                const unsigned int seq = r ? ts_old - 8u + (unsigned int)__popc(r) : (tkrow[(unsigned int)u] >> 8);
                tkrow[(unsigned int)u] = (seq << 8) | a;
              }
            }
          }
        }
        wave_lds_order();
      }
    };
    auto repack_pass = [&](unsigned int tkov) {
      {
        // the packed words again, from the (seq, age) words the pass left in `tkey`; the fresh sequence numbers;
        // the flags of the next slot: an entry 7 or more behind keeps its quad on this path
        if (ul < (unsigned int)PC) tsrow[ul] = tkov;
        ovf = ovf || (ul < (unsigned int)PC && tkov >= (1u << 24) - 1u);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          unsigned int ncw[VPL], naw[VPL];
          bool keep = false;
#pragma unroll
          for (int j = 0; j < VPL; ++j) { ncw[j] = 0u; naw[j] = 0u; }
#pragma unroll DIRAL_WIDE_FLAG_UNROLL
          for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * w + cc;
            const int k = kbase + c;
            if (!(FULL || k < N)) continue;
            const global_ptr<const unsigned int> tkrow = uniform_ptr<const unsigned int>(p.tkey, (bR + k) * NV);
            const unsigned int tk_own = (unsigned int)__builtin_amdgcn_readlane((int)tkov, c);
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
              const int u = lane + 64 * j;
              if (FULL || u < N) {
                const unsigned int wk = tkrow[(unsigned int)u];
                const unsigned int seqf = wk >> 8, lagf = tk_own - seqf;
                ncw[j] |= ((seqf != 0u && lagf <= 7u) ? ((0xffu << lagf) & 0xffu) : 0u) << (8 * cc);
                naw[j] |= (wk & 255u) << (8 * cc);
                keep = keep || (seqf != 0u && lagf >= 7u);
              }
            }
          }
          const bool anyk = __ballot(keep) != 0ull;
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            if (FULL || lane + 64 * j < N) {
              tcrow[(unsigned int)(w * NV) + ul + 64u * j] = ncw[j];
              tarow[(unsigned int)(w * NV) + ul + 64u * j] = naw[j];
            }
          }
          if (lane == 0) g_told[qrow + w] = anyk ? 1u : 0u;
          if (anyk && lane == 0) s_slow[0] = 1u;
        }
      }
    };
#if !DIRAL_WIDE_FUSED_FLAGGED
    unpack_pass();
#endif
    DIRAL_WCLOCK(tf1);
#define DIRAL_PASS_THERMO_FIRST false             // (a flagged pass: the codes do not reach - byte ranks, then 32-bit keys)
#define DIRAL_PASS_PACKED_IO DIRAL_WIDE_FUSED_FLAGGED
#include "step_wide_pass.inc"
#undef DIRAL_PASS_PACKED_IO
#undef DIRAL_PASS_THERMO_FIRST
    DIRAL_WCLOCK(tf2);
#if !DIRAL_WIDE_FUSED_FLAGGED
    repack_pass(tkov);
#endif
#ifdef DIRAL_TIMING
    DIRAL_WCLOCK(tf3);
    acc_load += tf1 - tf0; acc_merge += tf2 - tf1; acc_fin += tf3 - tf2;
#endif
      }
  } else {
#pragma unroll 1
  for (int pch = 0; pch < CPW / PC; ++pch) {
    const int kbase = wave * CPW + pch * PC;
    if (kbase >= NRows) break;
    if (EXTRA && RICH && p.notab) break;         // no piggybacked tables (test_env.py:138-139, 231-238): nothing to stamp, merge or observe
#define DIRAL_PASS_THERMO_FIRST true
#define DIRAL_PASS_PACKED_IO 0
#include "step_wide_pass.inc"
#undef DIRAL_PASS_PACKED_IO
#undef DIRAL_PASS_THERMO_FIRST
    if (!thermo && lane == 0) s_slow[0] = 1u;    // (the pass left the codes: byte ranks or 32-bit keys)
#ifdef DIRAL_TIMING
    DIRAL_WCLOCK(tc3);
    acc_load += tc1 - tc0; acc_merge += tc2 - tc1; acc_fin += tc3 - tc2;
#endif
  }
  }
  if (__ballot(ovf) != 0ull) {                     // (wave-uniform branch around the late-bound load)
    uint32_t* const errp = ((LateFastArgs)late_kernarg_base())->err;
    if (lane == 0) atomicOr(errp, kErrSeq);
  }
#ifdef DIRAL_TIMING
  if (lane == 0 && p.dbg) {      // synthetic stamps: accumulated load / merge / finalize time of all passes
    unsigned long long* d = p.dbg + ((size_t)b * WAVES + wave) * 8;
    d[3] = t_p3; d[4] = t_p3 + acc_load; d[5] = d[4] + acc_merge; d[6] = d[5] + acc_fin;
  }
#endif
  if constexpr (REGCNT) {
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      if (mycnt[j]) atomicAdd(&s_cnt[lane + 64 * j], mycnt[j]);
  }
  // `obs[user][i]` of the reference step (test_env.py:143, 206, 228, 240), rebuilt from the gather sources P1 left in LDS.
  // It depends on nothing behind P1: every wave writes its share right behind its own P3, in front of the barrier that
  // waits for the slowest one.  16 bytes per lane, consecutive lanes on consecutive pieces of a row; branch-free - the
  // gather-source words of a piece's resources in flight together, then the sources' positions (the divergent form
  // paid two dependent LDS round trips per VALUE); the distance is |dx| outright when the env's positions allow it
  // (p1_fast: decided once per env, as in P1).
  bool chobs_done = false;
  if constexpr (RICH && DIRAL_WIDE_P4V2) {
    const LateRichArgs lr0 = (LateRichArgs)(late_kernarg_base() + kRichArgOffset);
    void* const chobs_out0 = lr0->chobs_out;
    constexpr int CV = OUT64 ? 2 : 4;
    if (chobs_out0 && (A % CV) == 0) {
      typedef typename std::conditional<OUT64, double, float>::type out_t;
      const bool dist_obs = !CH && !(EXTRA && p.design) && lr0->state_type == 2;
      out_t* const co = static_cast<out_t*>(chobs_out0) + bN * A;
      const int qpr = A / CV, total = N * qpr;
      const int du = THREADS / qpr, dq = THREADS - du * qpr;
      auto run = [&](auto fast_tag) {
        constexpr bool ABS = decltype(fast_tag)::value;
        int u = tid / qpr, qr = tid - u * qpr;
        for (int q = tid; q < total; q += THREADS) {
          const int i0 = qr * CV;
          const int a = s_act[u];
          const double xu = s_px[u];
          const unsigned int tx_bits = (unsigned int)(actw >> i0);
          const unsigned int sh = 8u * (unsigned int)(u >> 6);
          const mword_t* const mrow = s_mtab + (u & 63) + i0 * MT;
          unsigned int srcv[CV];
#pragma unroll
          for (int d = 0; d < CV; ++d) srcv[d] = (unsigned int)mrow[d * MT];
          double xv[CV];
#pragma unroll
          for (int d = 0; d < CV; ++d) { srcv[d] = (srcv[d] >> sh) & 255u; xv[d] = s_px[srcv[d]]; }
          out_t o[CV];
          // (the value is selected in the OUTPUT type: float32(d) for the one float64 subtraction, then 100000 / 1 / 0 as
          // float32 constants - the same bits as float32 of the float64 selection, half the select instructions)
          const unsigned int zm = (~tx_bits) | ((unsigned int)(a - i0) < (unsigned int)CV ? 1u << (a - i0) : 0u);   // bit d: obs[u][i0 + d] = 0
#pragma unroll
          for (int d = 0; d < CV; ++d) {
            double dd;
            if constexpr (ABS) dd = __builtin_fabs(xu - xv[d]);
            else dd = fast_dist<true>(xv[d], 0.0, xu, 0.0);
            out_t val = (out_t)dd;
            val = (int)srcv[d] == u ? (out_t)100000.0 : val;                                    // network.py:385
            val = dist_obs ? val : (out_t)1.0;
            o[d] = ((zm >> d) & 1u) ? (out_t)0.0 : val;
          }
          if constexpr (OUT64) stream_store2(co + 2 * q, make_double2(o[0], o[1]));
          else stream_store4(co + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
          u += du; qr += dq;
          if (qr >= qpr) { qr -= qpr; u += 1; }
        }
      };
      if (p1_fast) run(std::true_type{});
      else run(std::false_type{});
      chobs_done = true;
    }
  }
  __syncthreads();
  // neighbours counted per viewer (network.py:497-501 `count`) = the row sum of its histogram
  // P4V2: where the plain state writer runs, every wave counts the NPAD / WAVES viewers whose rows it writes itself and
  // leaves count and 1 / n (one IEEE division: the value the host's table holds) in LDS for its own lanes - no second
  // barrier, no table load whose s_waitcnt vmcnt would wait for the streaming stores in front of it
  constexpr int RPW = NPAD / WAVES;              // state rows a wave writes
  bool rows_by_wave = false;
  if constexpr (!REGCNT && DIRAL_WIDE_P4V2) {
    bool plain_writer = true;
    if constexpr (RICH) plain_writer = ((LateRichArgs)(late_kernarg_base() + kRichArgOffset))->plain_state != 0;
    rows_by_wave = plain_writer && ((LateFastArgs)late_kernarg_base())->state_out != nullptr;
    if (rows_by_wave) {
      if (lane < RPW) {
        const int u = wave * RPW + lane;
        unsigned int n = 0u;
        for (int q = 0; q < (K + 1) / 2; ++q) { const unsigned int w = s_hist[u * KP + q]; n += (w & 0xffffu) + (w >> 16); }
        s_cnt[u] = n;
        if constexpr (!OUT64) reinterpret_cast<double*>(smem + lay.scratch)[u] = n ? 1.0 / (double)n : 0.0;
      }
      wave_lds_order();
    }
  }
  if (!REGCNT && !rows_by_wave) {
    if (tid < NPAD) {
      unsigned int n = 0u;
      for (int q = 0; q < (K + 1) / 2; ++q) { const unsigned int w = s_hist[tid * KP + q]; n += (w & 0xffffu) + (w >> 16); }
      s_cnt[tid] = n;
      // 1 / n for the float32 state vector, fetched HERE and parked in the merge scratch (dead since the barrier above): in
      // P4 the table load sat between streaming stores, and a wave's s_waitcnt vmcnt for it also waits for every store
      // issued before it (step_wide_closure.inc found the same for its table words)
      if constexpr (!OUT64 && DIRAL_WIDE_INV_LDS) {
        const double* const it = ((LateFastArgs)late_kernarg_base())->inv_tab;
        reinterpret_cast<double*>(smem + lay.scratch)[tid] = it[n < 256u ? n : 0u];
      }
    }
    __syncthreads();
  }

  // ---- P4: metrics, done flag, state = [one-hot(action) (A) | histogram (K)] ------
  const unsigned long long late = late_kernarg_base();             // outputs and the RICH section layout: from here on
  const LateFastArgs lp = (LateFastArgs)late;
  void* const state_out = lp->state_out;
  if (tid == 0) {
    // the next launch's order: a slow env asks for a place among the first blocks (as step_fast64_body.inc)
    uint32_t* const flag_w = lp->slow_flag_w;
    if (flag_w) {
      unsigned int fl = 0u;
      if (s_slow[0]) {
        const unsigned int pos = atomicAdd(lp->slow_cnt_w, 1u);
        if (pos < (unsigned int)fast_slow_max(lp->B)) { lp->slow_list_w[pos] = (unsigned int)b; fl = 1u; }
      }
      flag_w[b] = fl;
      // the set the launch after the next builds: count AND flags emptied (step_fast64.hpp)
      uint32_t* const set_z = lp->slow_cnt_z;
      set_z[16 + fast_slow_max(lp->B) + b] = 0u;
      if (b == 0) *set_z = 0u;
    }
    uint8_t* const done_out = lp->done_out;
    if (done_out) {
      int dn = lp->done_now;                                      // (slot clock: see step_fast64.hpp)
      const long long* const td = lp->t_dev;
      if (td) dn = ((unsigned int)(lp->t + *td) % (unsigned int)lp->episode_interval) == (unsigned int)lp->episode_interval - 1u;
      done_out[b] = (uint8_t)dn;
    }
    double sr = 0.0, sp = 0.0, ss = 0.0, sc = 0.0;
    for (int w = 0; w < VPL; ++w) { sr += s_red[w * 4 + 0]; sp += s_red[w * 4 + 1]; ss += s_red[w * 4 + 2]; sc += s_red[w * 4 + 3]; }
    double* mt = lp->metrics + (size_t)b * DIRAL_M_COLUMNS;
    unsafeAtomicAdd(&mt[DIRAL_M_SLOTS], 1.0);          // (no-return hardware atomics: no load to wait for, see step_fast64.hpp)
    unsafeAtomicAdd(&mt[DIRAL_M_SUM_REWARD], sr);
    unsafeAtomicAdd(&mt[DIRAL_M_TX_SOLE], ss);
    unsafeAtomicAdd(&mt[DIRAL_M_TX_COLLIDED], sc);
    if (CH || (EXTRA && lp->prr)) { unsafeAtomicAdd(&mt[DIRAL_M_PRR_SUM], sp); unsafeAtomicAdd(&mt[DIRAL_M_PRR_CNT], ss + sc); }
  }
  if constexpr (RICH) {
    const RichParams rr = load_rich_args(late);
    // `obs[user][i]` of the reference step, rebuilt from the gather sources (see step_fast64.hpp)
    const bool dist_obs = !CH && !(EXTRA && p.design) && rr.state_type == 2;
    // (`actw`: the wave-uniform word of resources with a transmitter, built before P3)
    auto chv_row = [&](int u, int a, double xu, int i) -> double {
      if (a == i || ((actw >> i) & 1ull) == 0ull) return 0.0;
      if (!dist_obs) return 1.0;
      const int src = (int)(((unsigned int)s_mtab[i * MT + (u & 63)] >> (8 * (u >> 6))) & 255u);
      if (src == u) return 100000.0;                                          // network.py:385
      return fast_dist<true>(s_px[src], 0.0, xu, 0.0);
    };
    auto chv = [&](int u, int i) -> double { return chv_row(u, s_act[u], s_px[u], i); };
    if (rr.chobs_out && !chobs_done) {
      // 16 bytes per lane, consecutive lanes on consecutive pieces of a row; the per-row values
      // (action, position) are loaded once per piece
      constexpr int CV = OUT64 ? 2 : 4;
      typedef typename std::conditional<OUT64, double, float>::type out_t;
      out_t* const co = static_cast<out_t*>(rr.chobs_out) + bN * A;
      if ((A % CV) == 0) {
        const int qpr = A / CV, total = N * qpr;
        // (row, piece) advance incrementally: one integer division per thread, not one per store; the
        // transmitter bits of the piece's resources and the byte lane of the viewer's gather sources are
        // taken once per piece
        const int du = THREADS / qpr, dq = THREADS - du * qpr;
        int u = tid / qpr, qr = tid - u * qpr;
        for (int q = tid; q < total; q += THREADS) {
          const int i0 = qr * CV;
          const int a = s_act[u];
          const double xu = s_px[u];
          const unsigned int tx_bits = (unsigned int)(actw >> i0);
          const unsigned int sh = 8u * (unsigned int)(u >> 6);
          const mword_t* const mrow = s_mtab + (u & 63);
          auto piece = [&](int d) -> double {
            const int i = i0 + d;
            if (a == i || ((tx_bits >> d) & 1u) == 0u) return 0.0;
            if (!dist_obs) return 1.0;
            const int src = (int)(((unsigned int)mrow[i * MT] >> sh) & 255u);
            if (src == u) return 100000.0;                                    // network.py:385
            return fast_dist<true>(s_px[src], 0.0, xu, 0.0);
          };
          if constexpr (OUT64) stream_store2(co + 2 * q, make_double2(piece(0), piece(1)));
          else stream_store4(co + 4 * q, make_float4((float)piece(0), (float)piece(1), (float)piece(2), (float)piece(3)));
          u += du; qr += dq;
          if (qr >= qpr) { qr -= qpr; u += 1; }
        }
      } else {
        for (int e = tid; e < N * A; e += THREADS) {
          const int u = e / A;
          stream_store(co + e, (out_t)chv(u, e - u * A));
        }
      }
    }
    if (state_out && !rr.plain_state) {
      // the reward column is read back from rew_out (written in P2 by this workgroup, two
      // barriers ago; the host dispatches here only with rew_out set when the column exists):
      // 2 KB of LDS for it would cost the third workgroup per CU at N = 256
      rich_write_state<OUT64>(
          rr, pflags, N, A, K, pL, state_out, bN, tid, THREADS, [&](int u) { return s_act[u]; }, chv,
          [&](int u, int bin) {
            const unsigned int n = s_cnt[u];
            const unsigned int h = (s_hist[u * KP + (bin >> 1)] >> (16 * (bin & 1))) & 0xffffu;
            return n ? (double)h / (double)n : 0.0;
          },
          [&](int u) {
            if constexpr (OUT64) return static_cast<const double*>(lp->rew_out)[bN + u];
            else return (double)static_cast<const float*>(lp->rew_out)[bN + u];
          },
          [&](int u) { return s_npx[u]; }, [&](int) { return 0.0; }, [&](int u) { return rr.vel[bN + u]; });
    }
    // plain state vector next to the channel observation: the vectorised writer below
    if (!(state_out && rr.plain_state)) { DIRAL_WSTAMP(7); return; }
  }
  const int S = A + K;
  // who writes which rows: all threads interleaved over the env's rows, or (rows_by_wave) each wave the RPW rows it counted
  const int T4 = rows_by_wave ? 64 : THREADS, t4 = rows_by_wave ? lane : tid;
  const int row0 = rows_by_wave ? wave * RPW : 0;
  const int row1 = rows_by_wave ? (row0 + RPW < N ? row0 + RPW : N) : N;
  if constexpr (OUT64) {
    double* out = static_cast<double*>(state_out) + bN * S;
    if (((A | K) & 1) == 0) {
      const int q_per_row = S >> 1, total = row1 * q_per_row;
      // (row, piece) advance incrementally: one integer division per thread instead of one per store
      const int du = T4 / q_per_row, dq = T4 - du * q_per_row;
      int u = t4 / q_per_row, qr = t4 - u * q_per_row;
      u += row0;
      for (int q = row0 * q_per_row + t4; q < total; q += T4, u += du, qr += dq) {
        if (qr >= q_per_row) { qr -= q_per_row; u += 1; }
        const int s0 = qr << 1;
        double2 v;
        if (s0 < A) {
          const int a = s_act[u] - s0;
          v = make_double2(a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0);
        } else {
          const unsigned int n = s_cnt[u];
          const unsigned int hw = s_hist[u * KP + ((s0 - A) >> 1)];           // s0 - A is even: one word
          const double dn = (double)n;
          v = n ? make_double2((double)(hw & 0xffffu) / dn, (double)(hw >> 16) / dn) : make_double2(0.0, 0.0);   // network.py:501
        }
        reinterpret_cast<double2*>(out)[q] = v;
      }
    } else {
      for (int e = row0 * S + t4; e < row1 * S; e += T4) {
        const int u = e / S, s = e - u * S;
        double val;
        if (s < A) val = (s_act[u] == s) ? 1.0 : 0.0;
        else {
          const unsigned int n = s_cnt[u];
          val = n ? (double)((s_hist[u * KP + ((s - A) >> 1)] >> (16 * ((s - A) & 1))) & 0xffffu) / (double)n : 0.0;
        }
        out[e] = val;
      }
    }
  } else {
    float* out = static_cast<float*>(state_out) + bN * S;
    const double* const inv_tab = lp->inv_tab;
    if (((A | K) & 3) == 0) {
      const int q_per_row = S >> 2, total = row1 * q_per_row;
      const int du = T4 / q_per_row, dq = T4 - du * q_per_row;
      int u = t4 / q_per_row, qr = t4 - u * q_per_row;
      u += row0;
      for (int q = row0 * q_per_row + t4; q < total; q += T4, u += du, qr += dq) {
        if (qr >= q_per_row) { qr -= q_per_row; u += 1; }
        const int s0 = qr << 2;
        float4 v;
        if (s0 < A) {
          const int a = s_act[u] - s0;
          v = make_float4(a == 0 ? 1.f : 0.f, a == 1 ? 1.f : 0.f, a == 2 ? 1.f : 0.f, a == 3 ? 1.f : 0.f);
        } else {
          const unsigned int n = s_cnt[u];
          const unsigned int* hw = s_hist + u * KP + ((s0 - A) >> 1);        // s0 - A is a multiple of 4: two words
          const unsigned int h01 = hw[0], h23 = hw[1];
          // one table load instead of four IEEE divisions: exact, see step_fast64.hpp
          const double inv = (!REGCNT && DIRAL_WIDE_INV_LDS) ? reinterpret_cast<const double*>(smem + lay.scratch)[u] : inv_tab[n];
          v = make_float4((float)((double)(h01 & 0xffffu) * inv), (float)((double)(h01 >> 16) * inv),
                          (float)((double)(h23 & 0xffffu) * inv), (float)((double)(h23 >> 16) * inv));
        }
        reinterpret_cast<float4*>(out)[q] = v;
      }
    } else {
      for (int e = row0 * S + t4; e < row1 * S; e += T4) {
        const int u = e / S, s = e - u * S;
        float val;
        if (s < A) val = (s_act[u] == s) ? 1.f : 0.f;
        else {
          const unsigned int n = s_cnt[u];
          val = n ? __fdiv_rn((float)((s_hist[u * KP + ((s - A) >> 1)] >> (16 * ((s - A) & 1))) & 0xffffu), (float)n) : 0.f;
        }
        out[e] = val;
      }
    }
  }
  DIRAL_WSTAMP(7);
}

}  // namespace diral
