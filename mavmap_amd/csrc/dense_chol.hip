// dense_chol.hip — structured SPD solve of the reduced camera system on gfx950.
//
// Replaces what the reference gets from Ceres' SPARSE_SCHUR back end (CHOLMOD
// factorisation of the reduced camera matrix, reference
// src/base3d/bundle_adjustment.cc:555). Right-looking blocked Cholesky, NB = 64, every product on
// v_mfma_f64_16x16x4_f64:
//
//   diag    a 64x64 diagonal tile is factorised AND inverted (tile_potrf_inv_la: 16x16 blocks whose 16
//           pivot steps are one rank-1 matrix instruction each, 4x4 blocks scheduled with look-ahead);
//   trsm    A_ik <- A_ik L_kk^-T is then a 64x64x64 product against the explicit inverse;
//   update  trailing tiles C_ij -= A_ik A_jk^T; the work-group that owns tile (k+1, k+1) goes on to
//           factorise + invert it, so a panel step is ONE launch (two for large trailing matrices);
//   fronts  with a nested-dissection order the uncoupled leading parts are factorised concurrently
//           (blockIdx.z = front); their contributions to the separator go to shadow blocks that are merged
//           before the separator's own (dense) chain;
//   solve   the right-hand side rides along as an extra row block (forward substitution is free); the
//           backward substitution is one launch of flag-synchronised work-groups, one per tile row.
//
// At BA sizes the cost is the chain of dependent panel steps, not the flops (DESIGN.md sections 4 and 6).
#include "internal.h"
#include "lm_bodies.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <mutex>
#include <type_traits>

namespace mavba {

namespace {
constexpr int NB = 64;
constexpr int GLD = 66;  // LDS pitch (doubles): (2*row + k) % 32 distinct per 32-lane group, 16 B rows

typedef double d4 __attribute__((ext_vector_type(4)));

// 1/sqrt(d) to full FP64 precision: v_rsq_f64 seed + two Newton steps (no divide, no sqrt
// expansion on the factorisation's critical path).
[[maybe_unused]] __device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  y = y * (1.5 - hd * y * y);
  y = y * (1.5 - hd * y * y);
  return y;
}

// Same value with one third-order (Halley) correction of the hardware seed instead of two Newton
// steps: y (1 + h (1/2 + 3/8 h)), h = 1 - d y^2. Seed error e -> ~e^3, two dependent operations shorter.
__device__ __forceinline__ double rsqrt_finish(double y, double h) {
  return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
}
__device__ __forceinline__ double rsqrt_halley(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  return rsqrt_finish(y, __builtin_fma(-(d * y), y, 1.0));
}

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// 16x16 SPD block: Cholesky factor AND the inverse of the factor, 16 pivot steps, the whole block
// living in ONE wave's MFMA accumulator layout (reg r of lane l = element [(l >> 4) + 4 r][l & 15]).
//
// Right-looking on the full symmetric block: at step J the scaled row J (u = a_J / sqrt(a_JJ),
// = column J of L by symmetry) sits in the 16 lanes of group t = J & 3, register J >> 2 — exactly
// where v_mfma_f64_16x16x4_f64 expects the k = t slice of both of its operands. So the rank-1
// update  A -= u u^T  is ONE matrix instruction whose operands are each lane's own (masked) value:
// no shuffles, no LDS, no per-column FMAs. The same instruction with the scaled row J of the
// running inverse as second operand applies the row operations to the identity, so X = L^-1 is
// finished together with the factor:  X -= u x_J^T.
// In: acc = the block (both triangles). Out: xacc = L^-1 (lower). Returns false on a bad pivot.
template <int J>
__device__ __forceinline__ void potrf_inv16_step(d4& acc, d4& xacc, d4& xfin, int lane, bool& ok) {
  constexpr int T4 = J & 3, RR = J >> 2;
  const double piv = readlane_d(acc[RR], 16 * T4 + J);
  ok = ok && (piv > 0.0);  // recorded, not repaired: a bad pivot just propagates NaNs and the solve is rejected
  const double rs = rsqrt_halley(piv);
  const bool in_row = (lane >> 4) == T4;
  const bool live = in_row && (lane & 15) >= J;
  const double u = live ? acc[RR] * rs : 0.0;      // row J of L^T (zero left of the diagonal)
  const double xj = in_row ? xacc[RR] * rs : 0.0;  // row J of the inverse, scaled = final
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
  xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, xj, xacc, 0, 0, 0);
  // Row J of acc / xacc is garbage now, but never read again: later steps only touch rows > J.
  // The finished inverse row is collected on the side (off the matrix-core dependency chain).
  xfin[RR] = in_row ? xj : xfin[RR];
}
// (Tried and measured slower on gfx950, scripts/_dbg/tile_bench.hip: running the scalar pivot recurrence
// piv_{J+1} = a11 - (a10 rs_J)^2 one step ahead of the matrix instructions so that the rsqrt overlaps the
// MFMA latency - 4600 instead of 4200 cycles per block, with or without a pinned instruction order.)
__device__ __forceinline__ bool potrf_inv16(d4& acc, d4& xacc, int lane) {
  bool ok = true;
  const int li = lane & 15, lk = lane >> 4;
  d4 xfin = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) xacc[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
  potrf_inv16_step<0>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<1>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<2>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<3>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<4>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<5>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<6>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<7>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<8>(acc, xacc, xfin, lane, ok);  potrf_inv16_step<9>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<10>(acc, xacc, xfin, lane, ok); potrf_inv16_step<11>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<12>(acc, xacc, xfin, lane, ok); potrf_inv16_step<13>(acc, xacc, xfin, lane, ok);
  potrf_inv16_step<14>(acc, xacc, xfin, lane, ok); potrf_inv16_step<15>(acc, xacc, xfin, lane, ok);
  xacc = xfin;
  return ok;
}

// The same factorisation FOUR pivots at a time. Rows 4b..4b+3 of the block are register b of the four 16-lane
// groups, i.e. exactly the k = 0..3 slices of one matrix instruction's B operand. With D4 = the 4x4 diagonal
// block of those rows, D4 = L4 L4^T and X4 = L4^-1 (ten wave-uniform values, computed redundantly by every lane
// from ten v_readlane pairs):
//   U  = X4 * rows(acc)    one instruction: A operand = X4 placed in rows 0..3, B operand = register b as it is;
//                          U (4 x 16, = rows 4b..4b+3 of L^T) lands in register 0 of the result, already in
//                          operand position for the next instruction
//   Xn = X4 * rows(xacc)   the same for the running inverse (final rows 4b..4b+3 of L^-1)
//   acc  -= U^T U          one instruction, rank 4
//   xacc -= U^T Xn         one instruction
// Four dependent matrix instructions + one 4x4 scalar factorisation per FOUR pivots instead of two dependent
// matrix instructions + rsqrt chain per pivot (measured: the 16-pivot block 4200 -> ~2000 cycles).
// m[q] = 1 in the lane that supplies entry q of X4 (rows of the lower triangle: (0,0) (1,0) (1,1) (2,0) ... (3,3)) to the
// matrix instruction's A operand - lane (k << 4 | i) supplies X4[i][k] -, 0 elsewhere.
struct Pivot4Masks { double m[10]; };
__device__ __forceinline__ Pivot4Masks pivot4_masks(int lane) {
  const int li = lane & 15, lk = lane >> 4;
  Pivot4Masks mk;
  int q = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k <= i; ++k) mk.m[q++] = (li == i && lk == k) ? 1.0 : 0.0;
  return mk;
}
// The scalar part of a 4-pivot step: the 4x4 diagonal block D4 of rows 4B..4B+3 (register B of the accumulator layout) is
// fetched with ten v_readlane pairs, factorised and inverted redundantly by every lane, and X4 = L4^-1 is returned placed as
// the matrix instruction's A operand (lane (k << 4 | i) holds X4[i][k], 0 elsewhere).
// Round 5 (scripts/_dbg/pivot_bench.hip, profiles/r05_pivot_bench.txt): this part is ISSUE-bound - ~88 instructions at ~4.1
// ticks + four v_rsq_f64 at ~17 = 428 ticks per step -, and it ADDS to the step's matrix instructions (64 ticks of the
// shared FP64 pipe each, whether anything waits for them or not): 733 = 428 + 309 with four of them, 607 with two.
template <int B>
__device__ __forceinline__ double pivot4_operand(const d4& acc, const Pivot4Masks& mk) {
  constexpr int c0 = 4 * B;
  // D4[a][b], a >= b: register B of lane (a << 4 | c0 + b)
  // The first pivot goes first and its reciprocal square root is started before the other nine values are fetched: the
  // wave issues in order, so the 18 v_readlane that follow fill the latency of v_rsq_f64 and of the correction's
  // dependent operations (the empty asm only pins that order: the compiler would fetch all ten values first).
  const double d00 = readlane_d(acc[B], 0 * 16 + c0 + 0);
  const double y00 = __builtin_amdgcn_rsq(d00);
  int alo = __double2loint(acc[B]), ahi = __double2hiint(acc[B]);
  asm volatile("" : "+v"(alo), "+v"(ahi) : "v"(y00));
  auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(ahi, l), __builtin_amdgcn_readlane(alo, l)); };
  const double d10 = rl(1 * 16 + c0 + 0), d11 = rl(1 * 16 + c0 + 1);
  const double d20 = rl(2 * 16 + c0 + 0), d21 = rl(2 * 16 + c0 + 1), d22 = rl(2 * 16 + c0 + 2);
  const double d30 = rl(3 * 16 + c0 + 0), d31 = rl(3 * 16 + c0 + 1), d32 = rl(3 * 16 + c0 + 2), d33 = rl(3 * 16 + c0 + 3);
  // 4x4 Cholesky, reciprocal diagonal r_k = 1 / l_kk
  const double r0 = rsqrt_finish(y00, __builtin_fma(-(d00 * y00), y00, 1.0));
  const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
  const double t11 = __builtin_fma(-l10, l10, d11);
  const double r1 = rsqrt_halley(t11);
  const double l21 = __builtin_fma(-l20, l10, d21) * r1, l31 = __builtin_fma(-l30, l10, d31) * r1;
  const double t22 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, d22));
  const double r2 = rsqrt_halley(t22);
  const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, d32)) * r2;
  const double t33 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, d33)));
  const double r3 = rsqrt_halley(t33);
  // (no per-pivot test: a pivot <= 0 makes its reciprocal square root NaN - v_rsq_f64 of a negative number, 0 * inf in the
  // correction of a zero -, the mask multiply-adds below carry it into every lane's operand and from there into every later
  // pivot: the callers look at the LAST step's operand once)
  // X4 = L4^-1 (lower)
  const double x10 = -(l10 * r0) * r1;
  const double x21 = -(l21 * r1) * r2;
  const double x32 = -(l32 * r2) * r3;
  const double x20 = -__builtin_fma(l21, x10, l20 * r0) * r2;
  const double x31 = -__builtin_fma(l32, x21, l31 * r1) * r3;
  const double x30 = -__builtin_fma(l32, x20, __builtin_fma(l31, x10, l30 * r0)) * r3;
  // A operand: lane (k = lk, i = li) supplies X4[i][k] for i < 4, k <= i. Round 4: ten multiply-adds with the lane's 0 / 1
  // masks (at most one of them is 1) instead of a ladder of 64-bit selects - 25 v_cndmask_b32, several compares and two
  // exec-mask regions per four pivots. Exact (x * 1 + 0); a bad pivot's NaN reaches every lane, and the solve is rejected anyway.
  double xa = mk.m[0] * r0;
  xa = __builtin_fma(mk.m[1], x10, xa); xa = __builtin_fma(mk.m[2], r1, xa);
  xa = __builtin_fma(mk.m[3], x20, xa); xa = __builtin_fma(mk.m[4], x21, xa); xa = __builtin_fma(mk.m[5], r2, xa);
  xa = __builtin_fma(mk.m[6], x30, xa); xa = __builtin_fma(mk.m[7], x31, xa); xa = __builtin_fma(mk.m[8], x32, xa);
  xa = __builtin_fma(mk.m[9], r3, xa);
  return xa;
}
template <int B>
__device__ __forceinline__ void potrf_inv16_block4(d4& acc, d4& xacc, d4& xfin, int lane, bool& ok, const Pivot4Masks& mk) {
  const int li = lane & 15, lk = lane >> 4;
  constexpr int c0 = 4 * B;
  (void)ok;
  const double xa = pivot4_operand<B>(acc, mk);
  const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
  const double ba = li >= c0 ? acc[B] : 0.0;  // columns left of the block are stale (never needed again)
  const d4 U = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, ba, zero, 0, 0, 0);
  const d4 Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xacc[B], zero, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);  // (both issued back to back: the second one runs in the first one's result latency)
  const double u = li >= c0 + lk ? U[0] : 0.0;  // row m of L^T is zero left of column c0 + m (rounding residue otherwise)
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, u, acc, 0, 0, 0);
  xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, Xn[0], xacc, 0, 0, 0);
  xfin[B] = Xn[0];
}
__device__ __forceinline__ bool potrf_inv16_b4(d4& acc, d4& xacc, int lane) {
  bool ok = true;
  const int li = lane & 15, lk = lane >> 4;
  d4 xfin = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) xacc[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
  const Pivot4Masks mk = pivot4_masks(lane);
  potrf_inv16_block4<0>(acc, xacc, xfin, lane, ok, mk);
  potrf_inv16_block4<1>(acc, xacc, xfin, lane, ok, mk);
  potrf_inv16_block4<2>(acc, xacc, xfin, lane, ok, mk);
  potrf_inv16_block4<3>(acc, xacc, xfin, lane, ok, mk);
  xacc = xfin;
  const double last = readlane_d(xfin[3], 63);  // 1 / l_15,15: a number only if all sixteen pivots were positive
  return last == last;
}

// acc += A(16x16, row-major lda) * B^T  (NT)   or   A * B (NN), K = 16, operands in LDS.
// v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][k = l >> 4], B[k = l >> 4][l & 15];
// reg r of lane l holds D[(l >> 4) + 4 r][l & 15].
__device__ __forceinline__ d4 gemm16(const double* A, int lda, const double* B, int ldb, bool transB,
                                     double sign, d4 acc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4) {
    const double av = sign * A[li * lda + kk + lk];
    const double bv = transB ? B[li * ldb + kk + lk] : B[(kk + lk) * ldb + li];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ d4 load_d16(const double* C, int ldc, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = C[(lk + 4 * r) * ldc + li];
  return v;
}
__device__ __forceinline__ void store_d16(double* C, int ldc, d4 v, int lane) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(lk + 4 * r) * ldc + li] = v[r];
}

constexpr bool kPivot4 = true;  // four pivots per matrix-instruction step in the 16x16 diagonal blocks (false: one)
constexpr int kFuseBelow = 24;  // fuse the panel solve into the update when <= this many row blocks remain ...
constexpr int kFuseTasks = 256;  // ... and the step (all fronts) has at most this many tile updates
constexpr int kMaxBacksolveGroups = 2048;  // single-launch backward substitution up to this many tile rows

// Order LDS traffic inside ONE wave: a block written by some lanes is read back by other lanes of
// the same wave. The hardware executes a wave's LDS instructions in order; this only stops the
// compiler from moving the loads above the stores.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Look-ahead variant of tile_potrf_inv: same result in Ti (L^-1, lower; upper zero), two LDS tiles,
// no extra scratch, 6 work-group barriers instead of 18.
//
// The critical chain of a blocked tile factorisation is  potrf(D_c) -> L_{c+1,c} -> D_{c+1} -> potrf.
// Wave 0 walks exactly that chain; everything else (the rest of panel c, the trailing update by
// panel c, the off-diagonal blocks of the inverse) is done by waves 1..3 WHILE wave 0 is inside the
// next 16-pivot factorisation, each wave re-deriving the few panel blocks it needs instead of waiting
// for their owner (a 16^3 product costs ~4 matrix instructions, a barrier costs the whole potrf).
//
// LDS use: T lower blocks = the matrix (column c stays raw, blocks right of it are updated in place
// by their owner wave); T upper blocks = per-wave scratch; Ti diagonal + lower = the inverse;
// Ti upper block (c, j) = L_jc while the factorisation runs (zeroed at the end).
struct NoPhaseHook { __device__ __forceinline__ void operator()(int) const {} };
// `hook(ph)`: called by every thread right after the barrier that ends phase ph (1..4): lets the caller slip a flag poll /
// a prefetch into the factorisation (the persistent chain asks for its next tiles there).
// `mark(id)`: wave 0's way points for the cycle harness (scripts/_dbg/tile_bench.hip); nothing in the product.
struct NoMark { __device__ __forceinline__ void operator()(int) const {} };
// `phase0(stage)`: the caller's work inside phase 0, where only wave 0 is busy (the persistent chain finishes the panel
// solve and the diagonal update of the column there): stage 0 - wave 0, before it factorises D_0 (its own part of the
// update); stage 1 - wave 0, right before the first block of the new inverse is written into Ti (Ti may still be read
// as the PREVIOUS column's inverse by the other waves: wait for them); stage 2 - waves 1..3, instead of idling; stage 3 -
// waves 1..3 at the start of phase 1 (their own panel work there is short: more of the caller's update fits).
struct NoPhase0 { __device__ __forceinline__ void operator()(int) const {} };
template <class Hook = NoPhaseHook, class Mark = NoMark, class Phase0 = NoPhase0>
__device__ __forceinline__ bool tile_potrf_inv_la(double* T, double* Ti, int tid, Hook hook = Hook(), Mark mark = Mark(),
                                                  Phase0 phase0 = Phase0()) {
  const int wv = tid >> 6, lane = tid & 63;
  const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
  auto Tb = [&](int i, int j) { return T + (16 * i) * GLD + 16 * j; };
  auto Xb = [&](int i, int j) { return Ti + (16 * i) * GLD + 16 * j; };
  auto Lb = [&](int i, int j) { return Ti + (16 * j) * GLD + 16 * i; };  // L_ij (i > j) parked at Ti block (j, i)
  // P = B * Dinv^T -> dst
  auto panel = [&](const double* B, const double* Dinv, double* dst) {
    d4 a = gemm16(B, GLD, Dinv, GLD, true, 1.0, zero, lane);
    store_d16(dst, GLD, a, lane);
  };
  // C -= Pa * Pb^T (in place)
  auto downdate = [&](double* C, const double* Pa, const double* Pb) {
    d4 a = load_d16(C, GLD, lane);
    a = gemm16(Pa, GLD, Pb, GLD, true, -1.0, a, lane);
    store_d16(C, GLD, a, lane);
  };
  // factor D_c = C - P P^T in registers, park its inverse in Ti(c, c)
  auto chain = [&](double* C, const double* P, int c) -> bool {
    d4 a = load_d16(C, GLD, lane), x;
    if (P) a = gemm16(P, GLD, P, GLD, true, -1.0, a, lane);
    mark(4 * c + 1);
    // Padding (every front of the elimination tree ends on a tile boundary: unit diagonal, zero elsewhere - and stays so
    // under the updates, whose panel rows are exact zeros): a block that IS the identity is its own factor and inverse.
    // One wave-wide compare instead of 16 dependent pivots (3 500 cycles); C3: ~6 such blocks on the critical path.
    {
      const int li = lane & 15, lk = lane >> 4;
      bool ident = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) ident = ident && a[r] == ((lk + 4 * r == li) ? 1.0 : 0.0);
      if (__builtin_amdgcn_ballot_w64(!ident) == 0) {
        if (c == 0) phase0(1);
        store_d16(Xb(c, c), GLD, a, lane);
        mark(4 * c + 2);
        return true;
      }
    }
    const bool ok = kPivot4 ? potrf_inv16_b4(a, x, lane) : potrf_inv16(a, x, lane);
    mark(4 * c + 2);
    if (c == 0) phase0(1);
    store_d16(Xb(c, c), GLD, x, lane);
    return ok;
  };
  bool ok = true;
  // ---- phase 0: D_0
  if (wv == 0) { phase0(0); ok = chain(Tb(0, 0), nullptr, 0); }
  else phase0(2);
  __syncthreads();
  hook(0);
  // ---- phase 1: panel 0; wave 0 goes on to D_1
  if (wv != 0) phase0(3);
  if (wv == 0) {
    mark(3);
    panel(Tb(1, 0), Xb(0, 0), Lb(1, 0));
    wave_lds_sync();
    mark(4);
    ok = chain(Tb(1, 1), Lb(1, 0), 1) && ok;
  } else if (wv == 1) {
    panel(Tb(2, 0), Xb(0, 0), Lb(2, 0));
    panel(Tb(1, 0), Xb(0, 0), Tb(0, 1));
    wave_lds_sync();
    downdate(Tb(2, 1), Lb(2, 0), Tb(0, 1));
    downdate(Tb(2, 2), Lb(2, 0), Lb(2, 0));
  } else if (wv == 2) {
    panel(Tb(3, 0), Xb(0, 0), Lb(3, 0));
    panel(Tb(1, 0), Xb(0, 0), Tb(0, 2));
    wave_lds_sync();
    downdate(Tb(3, 1), Lb(3, 0), Tb(0, 2));
    downdate(Tb(3, 3), Lb(3, 0), Lb(3, 0));
  } else {
    panel(Tb(3, 0), Xb(0, 0), Tb(1, 2));
    panel(Tb(2, 0), Xb(0, 0), Tb(1, 3));
    wave_lds_sync();
    downdate(Tb(3, 2), Tb(1, 2), Tb(1, 3));
  }
  __syncthreads();
  hook(1);
  // ---- phase 2: panel 1; wave 0 goes on to D_2; wave 3 starts on the inverse
  if (wv == 0) {
    mark(7);
    panel(Tb(2, 1), Xb(1, 1), Lb(2, 1));
    wave_lds_sync();
    mark(8);
    ok = chain(Tb(2, 2), Lb(2, 1), 2) && ok;
  } else if (wv == 1) {
    panel(Tb(3, 1), Xb(1, 1), Lb(3, 1));
    panel(Tb(2, 1), Xb(1, 1), Tb(0, 1));
    wave_lds_sync();
    downdate(Tb(3, 2), Lb(3, 1), Tb(0, 1));
  } else if (wv == 2) {
    panel(Tb(3, 1), Xb(1, 1), Tb(0, 2));
    wave_lds_sync();
    downdate(Tb(3, 3), Tb(0, 2), Tb(0, 2));
  } else {
    // X_10 = -Dinv_1 (L_10 Dinv_0)
    d4 m = gemm16(Lb(1, 0), GLD, Xb(0, 0), GLD, false, 1.0, zero, lane);
    store_d16(Tb(1, 2), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(1, 1), GLD, Tb(1, 2), GLD, false, -1.0, zero, lane);
    store_d16(Xb(1, 0), GLD, m, lane);
  }
  __syncthreads();
  hook(2);
  // ---- phase 3: panel 2; wave 0 goes on to D_3
  if (wv == 0) {
    mark(11);
    panel(Tb(3, 2), Xb(2, 2), Lb(3, 2));
    wave_lds_sync();
    mark(12);
    ok = chain(Tb(3, 3), Lb(3, 2), 3) && ok;
  } else if (wv == 1) {
    // X_21 = -Dinv_2 (L_21 Dinv_1)
    d4 m = gemm16(Lb(2, 1), GLD, Xb(1, 1), GLD, false, 1.0, zero, lane);
    store_d16(Tb(0, 1), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(2, 2), GLD, Tb(0, 1), GLD, false, -1.0, zero, lane);
    store_d16(Xb(2, 1), GLD, m, lane);
  } else if (wv == 2) {
    // X_20 = -Dinv_2 (L_20 X_00 + L_21 X_10)
    d4 m = gemm16(Lb(2, 0), GLD, Xb(0, 0), GLD, false, 1.0, zero, lane);
    m = gemm16(Lb(2, 1), GLD, Xb(1, 0), GLD, false, 1.0, m, lane);
    store_d16(Tb(0, 2), GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(2, 2), GLD, Tb(0, 2), GLD, false, -1.0, zero, lane);
    store_d16(Xb(2, 0), GLD, m, lane);
  } else {
    // (idle otherwise) the parts of the last block row's sums that do not need X_2j: L_30 X_00 + L_31 X_10, L_31 X_11
    d4 m = gemm16(Lb(3, 0), GLD, Xb(0, 0), GLD, false, 1.0, zero, lane);
    m = gemm16(Lb(3, 1), GLD, Xb(1, 0), GLD, false, 1.0, m, lane);
    store_d16(Tb(0, 3), GLD, m, lane);
    m = gemm16(Lb(3, 1), GLD, Xb(1, 1), GLD, false, 1.0, zero, lane);
    store_d16(Tb(1, 3), GLD, m, lane);
  }
  __syncthreads();
  hook(3);
  // ---- phase 4: last block row of the inverse, X_3j = -Dinv_3 sum_{k=j}^{2} L_3k X_kj
  mark(15);
  if (wv < 3) {
    const int j = 2 - wv;  // wave 0: X_32, wave 1: X_31, wave 2: X_30
    double* scr = wv == 0 ? Tb(1, 2) : (wv == 1 ? Tb(0, 1) : Tb(0, 2));
    d4 m = j == 2 ? zero : load_d16(j == 1 ? Tb(1, 3) : Tb(0, 3), GLD, lane);  // (wave 3's phase-3 partial sums)
    m = gemm16(Lb(3, 2), GLD, Xb(2, j), GLD, false, 1.0, m, lane);
    store_d16(scr, GLD, m, lane);
    wave_lds_sync();
    m = gemm16(Xb(3, 3), GLD, scr, GLD, false, -1.0, zero, lane);
    store_d16(Xb(3, j), GLD, m, lane);
  }
  __syncthreads();
  mark(16);
  // ---- the parked factor blocks leave the upper triangle of Ti
  for (int e = tid; e < 6 * 256; e += 256) {
    const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
    const int bi = b < 3 ? 0 : (b < 5 ? 1 : 2), bj = b < 3 ? b + 1 : (b < 5 ? b - 1 : 3);
    Ti[(16 * bi + r) * GLD + 16 * bj + c] = 0.0;
  }
  __syncthreads();
  return ok;
}

// ---- the SYSTOLIC tile factor + inverse (round 5) --------------------------------------------------------------------
// Same contract as tile_potrf_inv_la (T: the SPD tile, lower 16x16 blocks + full diagonal blocks, destroyed; Ti <- L^-1,
// lower, upper part zero), different organisation. scripts/_dbg/pivot_bench.hip showed why the look-ahead variant's 16-pivot
// block costs 2 950 ticks on wave 0 and a whole tile 17 800: the wave that walks the pivots is ISSUE-bound (428 ticks of
// scalar work per four pivots) and every matrix instruction it issues - also the two per step that only carry the running
// inverse, and the panel / update products between the blocks - adds 64 ticks of the shared FP64 pipe on top. So here the
// wave that holds the current diagonal block does NOTHING but the pivots (scalar part, U = X4 rows, acc -= U^T U: 607 ticks
// per four pivots) and publishes X4 (as the matrix instruction's A operand) and the masked U rows; every other block of
// the tile lives in the registers of ONE of the other waves for the whole factorisation and receives the same row
// operations there, one step behind:
//   * block (w, j), j < w, of the tile is held TRANSPOSED by wave w (its rows are then the pivot columns, i.e. the k
//     slices of the instruction's B operand): Lt = X4 * rows  finishes four rows of L_wj^T - already in operand position -,
//     Bt -= u^T Lt eliminates them from the rows below, and  (w, j')^T -= Lt_j'^T Lt_w,  D_w -= Lt_w^T Lt_w  are the
//     trailing updates, rank 4 per step, straight from registers (the other row's Lt arrives through LDS);
//   * wave c + 1 therefore holds the NEXT diagonal block complete ~300 ticks after the last pivot operand of block c is
//     out, and walks its pivots itself: the pivot role rotates 0 -> 1 -> 2 -> 3, no block ever moves;
//   * the inverse is the identity under the same row operations: X_cj <- X4 * rows (final), X_cj -= u^T Xn inside the
//     pivot block row (wave 2 for row 0, wave 0 - idle after its pivots - for rows 1 and 2, wave 1 for row 3),
//     X_ij -= Lt_i^T Xn_j below it; the below-parts are caught up from the published operands when their wave falls idle.
// Hand-offs go through a mailbox that aliases T (dead once every wave has its blocks): 16-double X4 records, 64-double
// records for u, Lt, Xn, one monotone progress counter per producer. A wave's LDS accesses execute in order, so a record
// followed by its counter needs no wait in between. Every wait is bounded: a time-out marks the tile failed (never a hang).
namespace systile {
// offsets in doubles. The counters sit in row 0 of T's upper block (0, 1) - no wave loads the upper blocks, so they can be
// cleared BEFORE the barrier behind the block loads -, the records fill T from double 128 to its end (4224).
constexpr int kFlags = 16, kXA = 128, kUU = kXA + 256, kLT = kUU + 768, kXN = kLT + 1536;
constexpr int F_XA = 0, F_U = 1, F_ROW = 1 /* + w, w = 1..3 */, F_FIN = 5, F_BAD = 6, F_DEAD = 7;
constexpr int F_PROW = 8 /* + w: row block w of the chain's panel tile is in LDS */, F_PST = 12 /* + w: ... and drained to memory */;
constexpr int kNumFlags = 16;
constexpr int kSpin = 1 << 16;
static_assert(kXN + 1536 <= NB * GLD && kFlags + kNumFlags / 2 <= 64, "the mailbox must fit the tile it aliases");
__host__ __device__ constexpr int pair_of(int c, int i) { return c == 0 ? i - 1 : (c == 1 ? i + 1 : 5); }
__host__ __device__ constexpr int q_of(int c, int j) { return c * (c + 1) / 2 + j; }

// (explicit LDS pointers: through generic ones every mailbox access became a system-scope FLAT access with a full drain
// behind it - 1 050 instead of 607 ticks per pivot step. Plain ds accesses, kept in program order by compiler barriers; the
// LDS executes one wave's accesses in order.)
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
__device__ __forceinline__ void order() { asm volatile("" ::: "memory"); }
struct Box {
  lds_f64* m;
  lds_i32* f;
  int lane, li, lk;
  __device__ __forceinline__ void wait(int flag, int v) const {
    int spin = 0;
    for (;;) {
      order();
      if (f[flag] >= v) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spin > kSpin || ((spin & 63) == 0 && f[F_DEAD] != 0)) { f[F_DEAD] = 1; break; }
    }
    order();
  }
  // (record + counter in ONE round trip: the LDS executes this wave's reads in order, so a record read issued behind the
  // counter read is at least as new as the counter - a dependent access costs ~130 ticks, two of them per operand were most
  // of a follower's step)
  __device__ __forceinline__ bool dead(int& spin) const {
    __builtin_amdgcn_s_sleep(1);
    if (++spin > kSpin || ((spin & 63) == 0 && f[F_DEAD] != 0)) { f[F_DEAD] = 1; return true; }
    return false;
  }
  // the pivot wave's operand of step `step` (= 4 phase + b) and, WITH_U, the masked rows u (us = 3 phase + b)
  template <bool WITH_U>
  __device__ __forceinline__ void fetch_pivot(int step, int us, double& xa, double& u) const {
    int spin = 0;
    for (;;) {
      order();
      const int have = WITH_U ? f[F_U] : f[F_XA];
      const double a = m[kXA + 16 * step + 4 * lk + (li & 3)];
      const double b = WITH_U ? m[kUU + 64 * us + lane] : 0.0;
      order();
      if (have >= (WITH_U ? us : step) + 1 || dead(spin)) { xa = li < 4 ? a : 0.0; u = b; return; }
    }
  }
  // the same reads without waiting: issued a step ahead, under the current step's matrix instructions; take_pivot falls
  // back to the blocking form when the counter was not there yet
  struct Pre { int have; double a, b; };
  template <bool WITH_U>
  __device__ __forceinline__ Pre pre_pivot(int step, int us) const {
    Pre p;
    order();
    p.have = WITH_U ? f[F_U] : f[F_XA];
    p.a = m[kXA + 16 * step + 4 * lk + (li & 3)];
    p.b = WITH_U ? m[kUU + 64 * us + lane] : 0.0;
    order();
    return p;
  }
  template <bool WITH_U>
  __device__ __forceinline__ void take_pivot(const Pre& p, int step, int us, double& xa, double& u) const {
    if (__builtin_amdgcn_readfirstlane(p.have) >= (WITH_U ? us : step) + 1) { xa = li < 4 ? p.a : 0.0; u = p.b; }
    else fetch_pivot<WITH_U>(step, us, xa, u);
  }
  __device__ __forceinline__ double fetch(int flag, int need, int off) const {
    int spin = 0;
    for (;;) {
      order();
      const int have = f[flag];
      const double v = m[off + lane];
      order();
      if (have >= need || dead(spin)) return v;
    }
  }
  __device__ __forceinline__ void signal(int flag, int v) const { order(); if (lane == 0) f[flag] = v; order(); }
  __device__ __forceinline__ void put(int off, double v) const { m[off + lane] = v; order(); }
  __device__ __forceinline__ double get(int off) const { order(); return m[off + lane]; }
  __device__ __forceinline__ void put_xa(int step, double xa) const { if (li < 4) m[kXA + 16 * step + 4 * lk + li] = xa; order(); }
  __device__ __forceinline__ double get_xa(int step) const {
    order();
    const double v = m[kXA + 16 * step + 4 * lk + (li & 3)];
    return li < 4 ? v : 0.0;
  }
};
__device__ __forceinline__ d4 mm(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ d4 zero4() { return (d4){0.0, 0.0, 0.0, 0.0}; }
__device__ __forceinline__ d4 ident4(int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
  return v;
}
// block (i, j) of the tile, transposed, in the accumulator layout: reg r of lane l = T[16 i + (l & 15)][16 j + (l >> 4) + 4 r]
__device__ __forceinline__ d4 load_d16t(const double* Tp, int i, int j, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = Tp[(16 * i + li) * GLD + 16 * j + lk + 4 * r];
  return v;
}

// pivot wave, block C, step B: operand out first (the next diagonal block's owner needs nothing else of the last step)
template <int C, int B>
__device__ __forceinline__ void pivot_step(const Box& bx, d4& acc, const Pivot4Masks& mk, bool& bad) {
  constexpr int c0 = 4 * B;
  const double xa = pivot4_operand<B>(acc, mk);
#ifndef MAVBA_SYS_EXP
#define MAVBA_SYS_EXP 0
#endif
  if constexpr (B < 3) {
    // (the record stores are issued BEHIND the matrix instruction they would otherwise delay: LDS and integer work runs in
    // its shadow, FP64 work does not. One counter for both records of a step: nobody needs the operand of steps 0..2 alone)
    const double ba = bx.li >= c0 ? acc[B] : 0.0;
    const d4 U = mm(xa, ba, zero4());
    __builtin_amdgcn_sched_barrier(0);
    if (MAVBA_SYS_EXP != 1) bx.put_xa(4 * C + B, xa);
    __builtin_amdgcn_sched_barrier(0);
    const double u = bx.li >= c0 + bx.lk ? U[0] : 0.0;
    acc = mm(-u, u, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (MAVBA_SYS_EXP != 1) {
      bx.put(kUU + 64 * (3 * C + B), u);
      bx.signal(F_U, 3 * C + B + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  } else {
    bx.put_xa(4 * C + B, xa);
    bx.signal(F_XA, 4 * C + B + 1);
    bad = bad || !(xa == xa);  // (a non-positive pivot anywhere in the block has turned the last operand into NaNs)
  }
}
template <int C>
__device__ __forceinline__ void pivot_block(const Box& bx, d4& acc, const Pivot4Masks& mk, bool& bad) {
  pivot_step<C, 0>(bx, acc, mk, bad); pivot_step<C, 1>(bx, acc, mk, bad);
  pivot_step<C, 2>(bx, acc, mk, bad); pivot_step<C, 3>(bx, acc, mk, bad);
}
// row owner W in phase C < W, step B: Bt[j] = block (W, j)^T for j < W, D = block (W, W)
template <int C, int B, int W>
__device__ __forceinline__ void row_step(const Box& bx, d4 (&Bt)[3], d4& D, Box::Pre& pre) {
  double xa, u;
  bx.template take_pivot<(B < 3)>(pre, 4 * C + B, 3 * C + B, xa, u);
  const d4 ltv = mm(xa, Bt[C][B], zero4());
  if constexpr (B < 3) pre = bx.template pre_pivot<(B + 1 < 3)>(4 * C + B + 1, 3 * C + B + 1);
  const double lt = ltv[0];
  bx.put(kLT + 64 * (4 * pair_of(C, W) + B), lt);
  bx.signal(F_ROW + W, 4 * C + B + 1);
  D = mm(-lt, lt, D);
  if constexpr (B < 3) Bt[C] = mm(-u, lt, Bt[C]);
#pragma unroll
  for (int j = C + 1; j < W; ++j) {
    const double ltj = bx.fetch(F_ROW + j, 4 * C + B + 1, kLT + 64 * (4 * pair_of(C, j) + B));
    Bt[j] = mm(-ltj, lt, Bt[j]);
  }
}
template <int C, int W>
__device__ __forceinline__ void row_phase(const Box& bx, d4 (&Bt)[3], d4& D) {
  Box::Pre pre = bx.template pre_pivot<true>(4 * C, 3 * C);
  row_step<C, 0, W>(bx, Bt, D, pre); row_step<C, 1, W>(bx, Bt, D, pre); row_step<C, 2, W>(bx, Bt, D, pre); row_step<C, 3, W>(bx, Bt, D, pre);
}
// block row C of the inverse under the pivot block's row operations: X[j] = X_Cj, j < NJ = C + 1; xf collects the final rows
template <int C, int B, int NJ, bool PUBLISH>
__device__ __forceinline__ void fin_step(const Box& bx, d4 (&X)[NJ], d4 (&xf)[NJ], double (&xn)[NJ], Box::Pre& pre) {
  double xa, u;
  bx.template take_pivot<(B < 3)>(pre, 4 * C + B, 3 * C + B, xa, u);
  d4 xv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) xv[j] = mm(xa, X[j][B], zero4());
  if constexpr (B < 3) pre = bx.template pre_pivot<(B + 1 < 3)>(4 * C + B + 1, 3 * C + B + 1);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    xn[j] = xv[j][0];
    xf[j][B] = xn[j];
    if constexpr (PUBLISH) bx.put(kXN + 64 * (4 * q_of(C, j) + B), xn[j]);
  }
  if constexpr (PUBLISH) bx.signal(F_FIN, 4 * C + B + 1);
  if constexpr (B < 3) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) X[j] = mm(-u, xn[j], X[j]);
  }
}
// X -= L_ic X_cj over the four steps of phase C, from the published operands (I = the row of X, J = the column block)
template <int C, int I, int J>
__device__ __forceinline__ void inv_trail_phase(const Box& bx, d4& X) {
  bx.wait(F_ROW + I, 4 * C + 4);
  bx.wait(F_FIN, 4 * C + 4);
  double l[4], x[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { l[b] = bx.get(kLT + 64 * (4 * pair_of(C, I) + b)); x[b] = bx.get(kXN + 64 * (4 * q_of(C, J) + b)); }
#pragma unroll
  for (int b = 0; b < 4; ++b) X = mm(-l[b], x[b], X);
}
}  // namespace systile

// `Panel`: the persistent chain's column prologue, fused in (round 5). With a sub-diagonal tile As (LDS) and the PREVIOUS
// column's inverse still in Ti, wave w first solves ITS 16 rows of the panel tile P = As Li^T - transposed, Pt = Li As_w^T, so
// that the result's registers are the k slices of a matrix instruction's operands -, leaves them in Cs (row-major, for the
// other waves and for the store to memory), downdates its own diagonal block from registers (D_w -= P_w P_w^T) and then its
// off-diagonal blocks as the rows above arrive in Cs. Wave 0 therefore starts on the pivots after 56 matrix instructions
// and no barrier; the waves with more blocks catch up under its scalar work.
// Wave 2 has nothing to do once it has walked its pivots (a quarter of the tile's time): `panel.prefetch_poll()` before them
// and `panel.prefetch()` behind them let the chain bring the NEXT column's tiles into LDS meanwhile.
// (`Panel`: hooks of the fused chain column of round 5 - ChainPanel, scripts/_dbg/pruned_r06.patch; with NoPanel every one of them
// is compiled out)
struct NoPanel { static constexpr bool enabled = false; };
template <class Mark = NoMark, int DEBUG_SOLO = 0, class Panel = NoPanel>
__device__ __forceinline__ bool tile_potrf_inv_sys(double* T, double* Ti, int tid, Mark mark = Mark(), Panel panel = Panel(), bool* stalled = nullptr) {
  using namespace systile;
  const int wv = tid >> 6, lane = tid & 63;
  // every wave takes its blocks (T is the mailbox from the barrier on); counters and the upper blocks of Ti are cleared meanwhile
  const Pivot4Masks mk = pivot4_masks(lane);
  Box bx{(lds_f64*)T, (lds_i32*)(T + kFlags), lane, lane & 15, lane >> 4};
  if (tid < kNumFlags) bx.f[tid] = 0;
  d4 Bt[3], D;
  D = load_d16(T + 16 * wv * GLD + 16 * wv, GLD, lane);
#pragma unroll
  for (int j = 0; j < 3; ++j) Bt[j] = j < wv ? load_d16t(T, wv, j, lane) : zero4();
  {
    double* z = Ti + (tid >> 4) * GLD + (tid & 15);  // element (r, c) of a 16x16 block
    z[16] = 0.0; z[32] = 0.0; z[48] = 0.0; z[16 * GLD + 32] = 0.0; z[16 * GLD + 48] = 0.0; z[32 * GLD + 48] = 0.0;
  }
  __syncthreads();
  bool with_panel = false;
  if constexpr (Panel::enabled) with_panel = panel.sub;
  if constexpr (Panel::enabled) if (with_panel) {
    // (measured and dropped: row block 0 of P - what the first pivots wait for - solved by the four waves side by side, one
    // column block each. Wave 0 then starts 1 700 ticks earlier, but the 8-16 extra matrix instructions come on top of the
    // followers' 72-104: wave 1 is late for ITS pivots and the column takes 11.1 instead of 10.2 us)
    const int li = lane & 15, lk = lane >> 4;
    d4 Pt[4];
    {
      const double* pa = panel.As + (16 * wv + li) * GLD + lk;
      double a[16], b0[4], b1[8], b2[12], b3[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[q] = pa[4 * q]; b3[q] = Ti[(48 + li) * GLD + lk + 4 * q]; }
#pragma unroll
      for (int q = 0; q < 12; ++q) b2[q] = Ti[(32 + li) * GLD + lk + 4 * q];
#pragma unroll
      for (int q = 0; q < 8; ++q) b1[q] = Ti[(16 + li) * GLD + lk + 4 * q];
#pragma unroll
      for (int q = 0; q < 4; ++q) b0[q] = Ti[li * GLD + lk + 4 * q];
#pragma unroll
      for (int n = 0; n < 4; ++n) Pt[n] = zero4();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q < 4) Pt[0] = mm(b0[q], a[q], Pt[0]);
        if (q < 8) Pt[1] = mm(b1[q], a[q], Pt[1]);
        if (q < 12) Pt[2] = mm(b2[q], a[q], Pt[2]);
        Pt[3] = mm(b3[q], a[q], Pt[3]);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) panel.Cs[(16 * wv + li) * GLD + 16 * n + lk + 4 * r] = Pt[n][r];
    order();
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) D = mm(-Pt[n][r], Pt[n][r], D);
    bx.signal(F_PROW + wv, 1);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < wv) {
        bx.wait(F_PROW + j, 1);
        double pj[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) pj[q] = panel.Cs[(16 * j + li) * GLD + 4 * q + lk];
#pragma unroll
        for (int q = 0; q < 16; ++q) Bt[j] = mm(-pj[q], Pt[q >> 2][q & 3], Bt[j]);
      }
    if (wv != 0) panel.store_rows(wv, lane);  // (wave 0: behind its pivots)
    mark(6);
  }
  // first store into Ti: every wave must be done with the previous column's inverse it still held (the panel solve above)
  auto ti_free = [&]() {
    if constexpr (Panel::enabled) if (with_panel) {
      bx.wait(F_PROW + 0, 1); bx.wait(F_PROW + 1, 1); bx.wait(F_PROW + 2, 1); bx.wait(F_PROW + 3, 1);
    }
  };
  // this wave's rows of the panel tile have reached memory
  auto panel_drained = [&]() {
    if constexpr (Panel::enabled) if (with_panel) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bx.signal(F_PST + wv, 1); }
  };
  bool bad = false;
  mark(0);
  if (DEBUG_SOLO) {  // (timing harness only: the pivot wave alone)
    if (wv == 0) { pivot_block<0>(bx, D, mk, bad); mark(1); }
  } else if (wv == 0) {
    pivot_block<0>(bx, D, mk, bad);
    if constexpr (Panel::enabled) if (with_panel) panel.store_rows(0, lane);
    mark(1);
    // rows 1 and 2 of the inverse. What phase 0 left below its pivot row first (all of it is published by now):
    d4 X1[2] = {zero4(), ident4(lane)}, X2[3] = {zero4(), zero4(), ident4(lane)};
    inv_trail_phase<0, 1, 0>(bx, X1[0]);
    inv_trail_phase<0, 2, 0>(bx, X2[0]);
    mark(2);
    {
      d4 xf[2] = {zero4(), zero4()};
      double xn[2];
      Box::Pre pre = bx.template pre_pivot<true>(4, 3);
      auto step = [&](auto bc) {
        constexpr int B = decltype(bc)::value;
        fin_step<1, B, 2, true>(bx, X1, xf, xn, pre);
        const double lt2 = bx.fetch(F_ROW + 2, 4 + B + 1, kLT + 64 * (4 * pair_of(1, 2) + B));  // row 2 below it: X_2j -= L_21 rows * Xn_1j
        X2[0] = mm(-lt2, xn[0], X2[0]);
        X2[1] = mm(-lt2, xn[1], X2[1]);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
      ti_free();
      store_d16(Ti + 16 * GLD, GLD, xf[0], lane);
      store_d16(Ti + 16 * GLD + 16, GLD, xf[1], lane);
      panel_drained();
        mark(3);
    }
    d4 X3b[2] = {zero4(), ident4(lane)};  // X_32, X_33: the second half of the inverse's last block row (wave 1 has the first)
    {
      d4 xf[3] = {zero4(), zero4(), zero4()};
      double xn[3];
      Box::Pre pre = bx.template pre_pivot<true>(8, 6);
      auto step = [&](auto bc) {
        constexpr int B = decltype(bc)::value;
        fin_step<2, B, 3, true>(bx, X2, xf, xn, pre);
        const double lt3 = bx.fetch(F_ROW + 3, 8 + B + 1, kLT + 64 * (4 * pair_of(2, 3) + B));
        X3b[0] = mm(-lt3, xn[2], X3b[0]);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
#pragma unroll
      for (int j = 0; j < 3; ++j) store_d16(Ti + 32 * GLD + 16 * j, GLD, xf[j], lane);
        mark(4);
    }
    {
      d4 xf[2] = {zero4(), zero4()};
      double xn[2];
      Box::Pre pre = bx.template pre_pivot<true>(12, 9);
      fin_step<3, 0, 2, false>(bx, X3b, xf, xn, pre); fin_step<3, 1, 2, false>(bx, X3b, xf, xn, pre);
      fin_step<3, 2, 2, false>(bx, X3b, xf, xn, pre); fin_step<3, 3, 2, false>(bx, X3b, xf, xn, pre);
      store_d16(Ti + 48 * GLD + 32, GLD, xf[0], lane);
      store_d16(Ti + 48 * GLD + 48, GLD, xf[1], lane);
    }
  } else if (wv == 1) {
    row_phase<0, 1>(bx, Bt, D);
    mark(1);
    pivot_block<1>(bx, D, mk, bad);
    mark(2);
    if constexpr (Panel::enabled) if (with_panel) {  // the panel tile goes public once all four row blocks are in memory
      panel_drained();
      bx.wait(F_PST + 0, 1); bx.wait(F_PST + 2, 1); bx.wait(F_PST + 3, 1);
      panel.publish(lane);
    }
    // X_30, X_31 of the inverse's last block row: what phases 0 and 1 left, then phase 2 as it is published, then the row's own phase
    d4 X3[2] = {zero4(), zero4()};
    inv_trail_phase<0, 3, 0>(bx, X3[0]);
    inv_trail_phase<1, 3, 0>(bx, X3[0]);
    inv_trail_phase<1, 3, 1>(bx, X3[1]);
    mark(3);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double lt3 = bx.fetch(F_ROW + 3, 8 + b + 1, kLT + 64 * (4 * pair_of(2, 3) + b));
      bx.wait(F_FIN, 8 + b + 1);
      double xn2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) xn2[j] = bx.get(kXN + 64 * (4 * q_of(2, j) + b));
#pragma unroll
      for (int j = 0; j < 2; ++j) X3[j] = mm(-lt3, xn2[j], X3[j]);
    }
    mark(4);
    d4 xf[2] = {zero4(), zero4()};
    double xn[2];
    Box::Pre pre = bx.template pre_pivot<true>(12, 9);
    fin_step<3, 0, 2, false>(bx, X3, xf, xn, pre); fin_step<3, 1, 2, false>(bx, X3, xf, xn, pre);
    fin_step<3, 2, 2, false>(bx, X3, xf, xn, pre); fin_step<3, 3, 2, false>(bx, X3, xf, xn, pre);
    ti_free();
#pragma unroll
    for (int j = 0; j < 2; ++j) store_d16(Ti + 48 * GLD + 16 * j, GLD, xf[j], lane);
    mark(5);
  } else if (wv == 2) {
    // row 2 of the tile and, while wave 0 walks the first block, row 0 of the inverse
    d4 X0[1] = {ident4(lane)}, xf[1] = {zero4()};
    double xn[1];
    Box::Pre pre = bx.template pre_pivot<true>(0, 0);
    auto step = [&](auto bc) {
      constexpr int B = decltype(bc)::value;
      double xa, u;
      bx.template take_pivot<(B < 3)>(pre, B, B, xa, u);
      const d4 ltv = mm(xa, Bt[0][B], zero4());
      const d4 xnv = mm(xa, X0[0][B], zero4());
      if constexpr (B < 3) pre = bx.template pre_pivot<(B + 1 < 3)>(B + 1, B + 1);
      const double lt = ltv[0];
      bx.put(kLT + 64 * (4 * pair_of(0, 2) + B), lt);
      bx.signal(F_ROW + 2, B + 1);
      xn[0] = xnv[0];
      xf[0][B] = xn[0];
      bx.put(kXN + 64 * (4 * q_of(0, 0) + B), xn[0]);
      bx.signal(F_FIN, B + 1);
      D = mm(-lt, lt, D);
      if constexpr (B < 3) {
        Bt[0] = mm(-u, lt, Bt[0]);
        X0[0] = mm(-u, xn[0], X0[0]);
      }
      Bt[1] = mm(-bx.fetch(F_ROW + 1, B + 1, kLT + 64 * (4 * pair_of(0, 1) + B)), lt, Bt[1]);
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    ti_free();
    store_d16(Ti, GLD, xf[0], lane);
    mark(1);
    row_phase<1, 2>(bx, Bt, D);
    panel_drained();
    mark(2);
    int next_state = 0;
    if constexpr (Panel::enabled) next_state = panel.prefetch_poll();  // (the flag loads travel under the pivots)
    pivot_block<2>(bx, D, mk, bad);
    if constexpr (Panel::enabled) {
      // As and Cs are read by the panel prologue and by the row stores only: every wave is past them (a formality by now)
      if (with_panel) { bx.wait(F_PST + 0, 1); bx.wait(F_PST + 1, 1); bx.wait(F_PST + 3, 1); }  // (drained = stored = read out of Cs; As was read before that)
      panel.prefetch(next_state, lane);
    }
    mark(3);
  } else {
    row_phase<0, 3>(bx, Bt, D);
    mark(1);
    row_phase<1, 3>(bx, Bt, D);
    panel_drained();
    mark(2);
    row_phase<2, 3>(bx, Bt, D);
    mark(3);
    pivot_block<3>(bx, D, mk, bad);
    mark(4);
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) bx.f[F_BAD] = 1;
  __syncthreads();
  const bool ok = bx.f[F_BAD] == 0 && bx.f[F_DEAD] == 0;
  if (stalled) *stalled = bx.f[F_DEAD] != 0;  // (a mailbox wait ran out: NOT "the matrix is not positive definite" - the caller decides)
  __syncthreads();  // (T - the mailbox - may be refilled by the caller from here on)
  return ok;
}

// The tile factorisation the kernels call (no hooks): the systolic one. (The look-ahead variant of rounds 2-4,
// tile_potrf_inv_la, stays as the persistent chain's tile for columns WITH a panel tile - see k_chol_persist.)
__device__ __forceinline__ bool tile_factor_inverse(double* T, double* Ti, int tid) { return tile_potrf_inv_sys(T, Ti, tid); }

// acc (2x2 MFMA tiles of the wave's 32x32 quadrant) = As(rows wr..) * Bs(rows wc..)^T, K = 64.
__device__ __forceinline__ void mfma_quadrant_nt(const double* As, const double* Bs, int wr, int wc,
                                                 int lane, d4 acc[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < NB; kk += 4) {
    const double a0 = As[(wr + li) * GLD + kk + lk];
    const double a1 = As[(wr + 16 + li) * GLD + kk + lk];
    const double b0 = Bs[(wc + li) * GLD + kk + lk];
    const double b1 = Bs[(wc + 16 + li) * GLD + kk + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// Tile store (internal.h): tile (i, k) of the matrix / the factor = store + slot[i * nb + k] * 4096, row-major with pitch NB.
__device__ __forceinline__ size_t tile_at(const int* __restrict__ slot, int nb, int i, int k) { return (size_t)slot[(size_t)i * nb + k] << 12; }
// 64x64 tile, global (leading dimension ld) -> LDS (pitch GLD), 256 threads, 16 B accesses.
__device__ __forceinline__ void load_tile(const double* __restrict__ G, size_t ld, double* S, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    const double2 v = *reinterpret_cast<const double2*>(G + (size_t)row * ld + c2);
    *reinterpret_cast<double2*>(S + row * GLD + c2) = v;
  }
}
__device__ __forceinline__ void store_tile(double* __restrict__ G, size_t ld, const double* S, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<double2*>(G + (size_t)row * ld + c2) = *reinterpret_cast<const double2*>(S + row * GLD + c2);
  }
}
}  // namespace

// Factor + invert the diagonal tiles listed in `tiles` (one work-group each): the first tile of every
// front. Block 0 also clears the backward substitution's flags for this solve.
__global__ void __launch_bounds__(256) k_chol_diag0(const double* __restrict__ M, const int* __restrict__ slot, int nb, const int* __restrict__ tiles,
                                                    double* __restrict__ inv, double* __restrict__ fail,
                                                    unsigned* __restrict__ flags, int nflags) {
  __shared__ __attribute__((aligned(16))) double T[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ti[NB * GLD];
  const int tid = threadIdx.x;
  const int t = tiles[blockIdx.x];
  if (blockIdx.x == 0)
    for (int f = tid; f < nflags; f += 256) flags[f] = 0u;
  load_tile(M + tile_at(slot, nb, t, t), NB, T, tid);
  __syncthreads();
  const bool ok = tile_factor_inverse(T, Ti, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(inv + (size_t)t * NB * NB, NB, Ti, tid);
}

// Panel solve, front blockIdx.z: row block blockIdx.x of the front's active list (the last one is the
// right-hand-side block):   A_ik <- A_ik L_kk^-T = A_ik (L_kk^-1)^T
__global__ void __launch_bounds__(256) k_chol_trsm(const double* __restrict__ M, double* __restrict__ Lout, const int* __restrict__ slot, int nb,
                                                   const CholFront* __restrict__ fronts,
                                                   const double* __restrict__ inv, const int* __restrict__ rows,
                                                   int aug) {
  const CholFront F = fronts[blockIdx.z];
  if ((int)blockIdx.x > F.na) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  const int tid = threadIdx.x;
  const int k = F.k;
  const int i = (int)blockIdx.x < F.na ? rows[F.act_off + blockIdx.x] : aug;  // active row block, or the right-hand side
  const size_t at = tile_at(slot, nb, i, k);
  const double* Ain = M + at;
  double* A = Lout + at;
  load_tile(Ain, NB, As, tid);
  load_tile(inv + (size_t)k * NB * NB, NB, Bs, tid);
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  d4 acc[2][2];
  mfma_quadrant_nt(As, Bs, wr, wc, lane, acc);
  // D layout of v_mfma_f64_16x16x4_f64: reg r of lane l holds D[(l >> 4) + 4 r][l & 15]
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        A[(wr + 16 * m + lk + 4 * r) * NB + wc + 16 * n + li] = acc[m][n][r];
}

// store the wave's 2x2 MFMA tiles (D layout) of a 64x64 product into an LDS tile (pitch GLD)
__device__ __forceinline__ void quadrant_to_lds(double* S, int wr, int wc, int lane, const d4 acc[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(wr + 16 * m + lk + 4 * r) * GLD + wc + 16 * n + li] = acc[m][n][r];
}

// Panel k, fused triangular solve + trailing update. Work-group (i, j), k < j <= i (i may be the
// right-hand-side block):
//   P_i = A_ik L_kk^-T,  P_j = A_jk L_kk^-T   (re-derived here from the raw panel tiles: two
//                                               extra 64^3 MFMA products instead of a launch)
//   tile (i, j) -= P_i P_j^T
// Column j == k+1 work-groups store P_i (the final L_ik, needed by the backward substitution);
// the owner of tile (k+1, k+1) then factorises + inverts it for the next panel.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_chol_update(double* __restrict__ M, double* __restrict__ Lout, const int* __restrict__ slot, int nb,
                                                     const CholFront* __restrict__ fronts,
                                                     double* __restrict__ inv, double* __restrict__ fail,
                                                     const int* __restrict__ rows, int aug,
                                                     double* __restrict__ shadow) {
  const CholFront F = fronts[blockIdx.z];
  const int na = F.na, k = F.k;
  if ((int)blockIdx.x >= na || (int)blockIdx.y > na) return;
  // only the row blocks whose envelope reaches panel k take part (act[0] is k + 1 when the front goes on)
  const int* act = rows + F.act_off;
  const int j = act[blockIdx.x], i = (int)blockIdx.y < na ? act[blockIdx.y] : aug;
  if (j > i) return;
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Cs[NB * GLD];
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  const int li = lane & 15, lk = lane >> 4;
  // the tile being updated (ancestor columns of a front with a shadow: that front's shadow block);
  // its loads are issued first so that their latency hides behind the products
  double* C = M + tile_at(slot, nb, i, j);
  size_t ldc = (size_t)NB;
  if (F.sh_off >= 0 && j >= F.sh_begin) {
    ldc = (size_t)(aug - F.sh_begin) * NB;
    C = shadow + F.sh_off + (size_t)(i - F.sh_begin) * NB * ldc + (size_t)(j - F.sh_begin) * NB;
  }
  d4 cin[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) cin[m][n][r] = C[(size_t)(wr + 16 * m + lk + 4 * r) * ldc + wc + 16 * n + li];
  d4 acc[2][2];
  if constexpr (FUSED) {
    load_tile(M + tile_at(slot, nb, i, k), NB, As, tid);
    load_tile(inv + (size_t)k * NB * NB, NB, Bs, tid);
    if (i != j) load_tile(M + tile_at(slot, nb, j, k), NB, Cs, tid);
    __syncthreads();
    mfma_quadrant_nt(As, Bs, wr, wc, lane, acc);      // P_i
    d4 accj[2][2];
    if (i != j) mfma_quadrant_nt(Cs, Bs, wr, wc, lane, accj);  // P_j
    __syncthreads();
    quadrant_to_lds(As, wr, wc, lane, acc);
    if (i != j) quadrant_to_lds(Cs, wr, wc, lane, accj);
    __syncthreads();
    // L_ik is final; it goes to the second matrix (the raw tile is still being read by the
    // other work-groups of this launch)
    if (blockIdx.x == 0) store_tile(Lout + tile_at(slot, nb, i, k), NB, As, tid);
  } else {
    // panel tiles were already solved by k_chol_trsm into the second matrix
    load_tile(Lout + tile_at(slot, nb, i, k), NB, As, tid);
    if (i != j) load_tile(Lout + tile_at(slot, nb, j, k), NB, Cs, tid);
    __syncthreads();
  }
  const double* Pj = (i != j) ? Cs : As;
  mfma_quadrant_nt(As, Pj, wr, wc, lane, acc);
  const bool next_diag = (blockIdx.x == 0 && blockIdx.y == 0 && F.factor_next);
  if (!next_diag) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          C[(size_t)(wr + 16 * m + lk + 4 * r) * ldc + wc + 16 * n + li] = cin[m][n][r] - acc[m][n][r];
    return;
  }
  // tile (k+1, k+1): keep the updated tile in LDS, factorise + invert it for the next panel
  __syncthreads();  // everyone is done reading As / Cs
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr + 16 * m + lk + 4 * r) * GLD + wc + 16 * n + li] = cin[m][n][r] - acc[m][n][r];
  __syncthreads();
  const bool ok = tile_factor_inverse(Cs, As, tid);
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  store_tile(inv + (size_t)(k + 1) * NB * NB, NB, As, tid);
}

// End of a tree level: M[i][j] += sum of the level's shadow blocks that cover tile (i, j) (i may be the
// right-hand-side row). A shadow with origin o covers tiles i, j >= o. grid = (nb - begin, nb - begin + 1).
__global__ void __launch_bounds__(256) k_chol_merge(double* __restrict__ M, const int* __restrict__ slot, int nb, const double* __restrict__ shadow,
                                                    const CholMerge* __restrict__ merges, int num_merges, int begin,
                                                    int aug) {
  const int j = begin + blockIdx.x, i = begin + blockIdx.y;
  if (j > i) return;
  const int sl = slot[(size_t)i * nb + j];
  if (sl < 0) return;  // (outside the envelope: no shadow block writes there either)
  double* C = M + ((size_t)sl << 12);
  for (int e = threadIdx.x; e < NB * NB / 2; e += 256) {
    const int row = e >> 5, c2 = (e & 31) * 2;
    double2 v = *reinterpret_cast<const double2*>(C + row * NB + c2);
    for (int q = 0; q < num_merges; ++q) {
      const int o = merges[q].sh_begin;
      if (j < o) continue;  // (i >= j >= o)
      const size_t lds = (size_t)(aug - o) * NB;
      const double* Sh = shadow + merges[q].sh_off + (size_t)(i - o) * NB * lds + (size_t)(j - o) * NB;
      const double2 w = *reinterpret_cast<const double2*>(Sh + (size_t)row * lds + c2);
      v.x += w.x; v.y += w.y;
    }
    *reinterpret_cast<double2*>(C + row * NB + c2) = v;
  }
}

// Backward substitution, tile k (fallback for very many tile rows, single segment only):
// y_k = L_kk^-T z_k; then z_j -= L_kj^T y_k for j < k.
// grid = k + 1: block k stores y_k, block j < k updates z_j (disjoint segments).
__global__ void __launch_bounds__(64) k_chol_backsolve(const double* __restrict__ M, const int* __restrict__ slot, int nb, int k, int first,
                                                       const double* __restrict__ inv,
                                                       double* __restrict__ z, double* __restrict__ y) {
  __shared__ double zk[NB];
  __shared__ double yk[NB];
  const int lane = threadIdx.x;
  zk[lane] = z[tile_at(slot, nb, nb, k) + lane];  // (z = the factor's store: the forward-substituted right-hand side is the first row of tile (nb, k))
  __syncthreads();
  const double* Li = inv + (size_t)k * NB * NB;  // L_kk^-1, lower, compact 64x64
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; m += 2) {
    acc0 += Li[m * NB + lane] * zk[m];
    acc1 += Li[(m + 1) * NB + lane] * zk[m + 1];
  }
  const double mine = acc0 + acc1;  // (L^-T z)_lane = sum_m Linv[m][lane] z_m
  const int j = first + blockIdx.x;  // L_kj is structurally zero left of `first`
  if (j == k) { y[k * NB + lane] = mine; return; }
  yk[lane] = mine;
  __syncthreads();
  const int sl = slot[(size_t)k * nb + j];
  if (sl < 0) return;  // (inside [first, k) but outside the envelope: structurally zero)
  const double* L = M + ((size_t)sl << 12);
  double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
  for (int m = 0; m < NB; m += 2) {
    a0 += L[m * NB + lane] * yk[m];
    a1 += L[(m + 1) * NB + lane] * yk[m + 1];
  }
  z[tile_at(slot, nb, nb, j) + lane] -= a0 + a1;
}
__global__ void k_scatter_y(int n, const int* __restrict__ scatter, const double* __restrict__ y, double* __restrict__ y_nat) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && scatter[t] >= 0) y_nat[scatter[t]] = y[t];
}

// Backward substitution L^T x = z in ONE launch: a work-group owns tile rows k,
//   x_k = L_kk^-T (z_k - sum_{i > k} L_ik^T x_i),
// and consumes the x_i in decreasing i as their owners publish them (flag per tile carrying the solve's epoch; the
// segments travel by system-scope write-through stores / loads, the flags are relaxed: no fences). Each of the 4 waves takes 16 of the 64 rows of every tile; the tile values
// are fetched BEFORE the wait, so a step of the chain is {flag + 64 values of x, 16 FMAs, two LDS
// reductions}, not a kernel launch. Tile (i, k) takes part iff k is inside row i's envelope for k's
// segment - uncoupled parts of a nested-dissection ordering therefore never wait for each other.
__global__ void __launch_bounds__(256) k_chol_backsolve_all(const double* __restrict__ L, const int* __restrict__ slot, int nb, int nseg,
                                                            const int* __restrict__ seg_of_tile,
                                                            const int* __restrict__ seg_first,
                                                            const double* __restrict__ inv,
                                                            double* y,
                                                            unsigned* flags, unsigned epoch, const int* __restrict__ scatter,
                                                            double* __restrict__ y_nat) {
  __shared__ double part[4][NB];
  __shared__ double xs[4][16];  // a wave's 16 values of the published segment it is consuming
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  // Work-group b owns tile rows nb-1-b, nb-1-b-G, ... (decreasing): every row it waits for belongs to a work-group
  // that reaches it without waiting for this one, so a resident grid (G <= 2 per CU) cannot dead-lock whatever the
  // dispatch order.
  for (int k = nb - 1 - (int)blockIdx.x; k >= 0; k -= (int)gridDim.x) {
  const int sk = seg_of_tile[k];
  double acc = (wv == 0) ? L[tile_at(slot, nb, nb, k) + lane] : 0.0;  // (right-hand side: first row of tile (nb, k))
  // this wave's 16 rows of L_kk^-1 (lower, compact 64 x 64)
  double li[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) li[q] = inv[(size_t)k * NB * NB + (size_t)(16 * wv + q) * NB + lane];
  int i = nb - 1;
  while (i > k && seg_first[i * nseg + sk] > k) --i;  // tiles outside the envelope are structurally zero
  double l[16];
  if (i > k) {
    const double* Lt = L + tile_at(slot, nb, i, k) + (16 * wv) * NB + lane;
#pragma unroll
    for (int q = 0; q < 16; ++q) l[q] = Lt[q * NB];
  }
  while (i > k) {
    int nx = i - 1;
    while (nx > k && seg_first[nx * nseg + sk] > k) --nx;
    double ln[16];
    if (nx > k) {
      const double* Lt = L + tile_at(slot, nb, nx, k) + (16 * wv) * NB + lane;
#pragma unroll
      for (int q = 0; q < 16; ++q) ln[q] = Lt[q * NB];
    }
    // hand-off: the owner wrote x_i with system-scope (write-through) stores, drained them and raised the flag; the
    // poll is relaxed and the values are read with system-scope loads - no fences (an agent-scope release / acquire
    // pair costs 3-8 us per hop here), placement independent
    while (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
    if (lane < 16) xs[wv][lane] = __hip_atomic_load(y + (size_t)i * NB + 16 * wv + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    wave_lds_sync();
    const double* xi = xs[wv];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      a0 = __builtin_fma(l[q], xi[q], a0);
      a1 = __builtin_fma(l[q + 1], xi[q + 1], a1);
    }
    acc -= a0 + a1;
    wave_lds_sync();  // xs[wv] is rewritten in the next round
#pragma unroll
    for (int q = 0; q < 16; ++q) l[q] = ln[q];
    i = nx;
  }
  part[wv][lane] = acc;
  __syncthreads();
  const double r = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];  // (z_k - sum)_lane
  __syncthreads();
  if (wv == 0) part[0][lane] = r;
  __syncthreads();
  // x_k[c] = sum_m Linv[m][c] r[m]; this wave's m range
  double b0 = 0.0, b1 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q += 2) {
    b0 = __builtin_fma(li[q], part[0][16 * wv + q], b0);
    b1 = __builtin_fma(li[q + 1], part[0][16 * wv + q + 1], b1);
  }
  __syncthreads();
  part[wv][lane] = b0 + b1;
  __syncthreads();
  if (wv == 0) {
    const double x = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    __hip_atomic_store(y + (size_t)k * NB + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&flags[k], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (scatter) {  // (read by later launches only)
      const int t = scatter[k * NB + lane];
      if (t >= 0) y_nat[t] = x;
    }
  }
  __syncthreads();  // `part` is reused by the next row
  }
}


// ===========================================================================================================
// Persistent forward factorisation: ONE launch for the whole factorisation + forward substitution.
//
// The launch-per-panel schedule above pays a launch + drain, one redundant 64^3 product and the tile loads on
// every one of the dependent panel steps. Here the work-groups stay resident (one per CU) and hand tiles to
// each other through memory:
//   * owner-computes, left-looking by tile: the owner of tile (i, j) applies the updates -P_ik P_jk^T of all
//     columns k that reach it in a FIXED order (by availability: schedule step of k, then k), then solves it
//     against L_jj^-1 and publishes it. No two work-groups ever write the same tile, so the concurrent fronts
//     of the elimination tree need no shadow blocks and no merge pass; results do not depend on timing.
//   * one CHAIN work-group per node of the tree walks the node's diagonal: L_jj^-1 never leaves its LDS between
//     column j and column j + 1 (panel solve of tile (j+1, j), its contribution to tile (j+1, j+1), tile factor +
//     inverse). The updates of those two tiles by earlier columns are applied by helper tasks (PRE) while the
//     chain is inside the previous tile factorisation.
//   * hand-off protocol (placement independent): payload tiles are written and read with system-scope
//     write-through accesses (sc0 sc1: they bypass the non-coherent per-XCD L2s and the per-CU L1), the
//     producer drains its stores (s_waitcnt vmcnt(0)) before a relaxed agent-scope flag store, the consumer
//     polls the flag with relaxed agent-scope loads. No fences. Flags carry the solve's epoch, so nothing has
//     to be cleared between solves.
//   * static schedule, built on the host (CholStructure::build): work-groups are partitioned by tree level and
//     every work-group runs its task list in an order that is topological for the whole task graph, so the
//     launch cannot dead-lock as long as all work-groups are resident (grid <= number of CUs, 1 work-group per
//     CU by LDS size). Every wait is bounded: on a time-out the launch raises the abort flag, adds 1e30 to
//     *fail and exits; the host then repeats the solve with the launch-per-panel schedule.
// ===========================================================================================================
namespace {
typedef int i4v __attribute__((ext_vector_type(4)));
constexpr int kCoherent = 17;  // buffer cache policy: sc0 | sc1
constexpr int kSpinLimit = 400000;  // polls of one wait (~0.3 s) before the launch gives up

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00020000);
}
// 64x64 tile, global (leading dimension ld) <-> LDS (pitch GLD), 256 threads, 16 B system-scope accesses
__device__ __forceinline__ void load_tile_coh(const double* G, size_t ld, double* S, int tid) {
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(G);
  i4v v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(((size_t)row * ld + c2) * 8), 0, kCoherent);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<i4v*>(S + row * GLD + c2) = v[q];
  }
}
__device__ __forceinline__ void store_tile_coh(double* G, size_t ld, const double* S, int tid) {
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(G);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    const i4v v = *reinterpret_cast<const i4v*>(S + row * GLD + c2);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(((size_t)row * ld + c2) * 8), 0, kCoherent);
  }
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the same tile traffic in two halves: global -> registers (loads stay in flight), registers -> LDS later
struct TileRegs { i4v v[8]; };
__device__ __forceinline__ void load_tile_regs(const double* G, size_t ld, TileRegs& R, int tid, bool coherent) {
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(G);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    const int off = (int)(((size_t)row * ld + c2) * 8);
    R.v[q] = coherent ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kCoherent) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  }
}
__device__ __forceinline__ void tile_regs_to_lds(const TileRegs& R, double* S, int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 5, c2 = (idx & 31) * 2;
    *reinterpret_cast<i4v*>(S + row * GLD + c2) = R.v[q];
  }
}
// Out = As * Li^T for a LOWER-triangular Li (the inverse of a diagonal factor tile): column block n of the product
// only takes the first 16 (n + 1) columns of As. Wave w: rows 32 (w >> 1) .. + 31, column blocks {0, 3} or {1, 2}
// (20 of the 16 x 4 k-steps each: balanced). 160 matrix instructions instead of 256, 40 per wave instead of 64.
// Everything is unrolled at compile time: all operand reads of the wave are issued first (one LDS latency), then the
// matrix instructions run back to back on four independent accumulators.
template <int N0, int N1>
__device__ __forceinline__ void trsm_tri_pair(const double* As, const double* Li, double* Out, int r0, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  constexpr int K0 = 4 * (N0 + 1), K1 = 4 * (N1 + 1);
  double a0[K1], a1[K1], b0[K0], b1[K1];
  const double* pa0 = As + (r0 + li) * GLD + lk;
  const double* pa1 = As + (r0 + 16 + li) * GLD + lk;
  const double* pb0 = Li + (16 * N0 + li) * GLD + lk;
  const double* pb1 = Li + (16 * N1 + li) * GLD + lk;
#pragma unroll
  for (int s = 0; s < K1; ++s) { a0[s] = pa0[4 * s]; a1[s] = pa1[4 * s]; b1[s] = pb1[4 * s]; }
#pragma unroll
  for (int s = 0; s < K0; ++s) b0[s] = pb0[4 * s];
  d4 c00 = (d4){0.0, 0.0, 0.0, 0.0}, c10 = c00, c01 = c00, c11 = c00;
#pragma unroll
  for (int s = 0; s < K1; ++s) {
    if (s < K0) {
      c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], c00, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b0[s], c10, 0, 0, 0);
    }
    c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b1[s], c01, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], c11, 0, 0, 0);
  }
  store_d16(Out + r0 * GLD + 16 * N0, GLD, c00, lane);
  store_d16(Out + (r0 + 16) * GLD + 16 * N0, GLD, c10, lane);
  store_d16(Out + r0 * GLD + 16 * N1, GLD, c01, lane);
  store_d16(Out + (r0 + 16) * GLD + 16 * N1, GLD, c11, lane);
}
__device__ __forceinline__ void trsm_lower_tri(const double* As, const double* Li, double* Out, int wv, int lane) {
  const int r0 = 32 * (wv >> 1);
  if (wv & 1) trsm_tri_pair<1, 2>(As, Li, Out, r0, lane);
  else trsm_tri_pair<0, 3>(As, Li, Out, r0, lane);
}
// ---- the chain's panel solve and diagonal update (D -= P P^T on the lower 16x16 blocks), cut so that they fit AROUND the
// tile factorisation (persistent chain) ----
// One 16x16 block of P = As Li^T: rows [16 rb, +16), column block N (Li lower triangular: 4 (N + 1) k-steps).
template <int N>
__device__ __forceinline__ d4 trsm_block16(const double* As, const double* Li, int rb, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  constexpr int K = 4 * (N + 1);
  const double* pa = As + (16 * rb + li) * GLD + lk;
  const double* pb = Li + (16 * N + li) * GLD + lk;
  double a[K], b[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { a[s] = pa[4 * s]; b[s] = pb[4 * s]; }
  d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < K; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
  return c;
}
// A whole 16-row block of P: the four column blocks on four independent accumulators (40 matrix instructions).
__device__ __forceinline__ void trsm_row16(const double* As, const double* Li, int rb, int lane, d4 (&c)[4]) {
  const int li = lane & 15, lk = lane >> 4;
  const double* pa = As + (16 * rb + li) * GLD + lk;
  double a[16], b0[4], b1[8], b2[12], b3[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) { a[s] = pa[4 * s]; b3[s] = Li[(48 + li) * GLD + lk + 4 * s]; }
#pragma unroll
  for (int s = 0; s < 12; ++s) b2[s] = Li[(32 + li) * GLD + lk + 4 * s];
#pragma unroll
  for (int s = 0; s < 8; ++s) b1[s] = Li[(16 + li) * GLD + lk + 4 * s];
#pragma unroll
  for (int s = 0; s < 4; ++s) b0[s] = Li[li * GLD + lk + 4 * s];
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s < 4) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b0[s], c[0], 0, 0, 0);
    if (s < 8) c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b1[s], c[1], 0, 0, 0);
    if (s < 12) c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b2[s], c[2], 0, 0, 0);
    c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b3[s], c[3], 0, 0, 0);
  }
}
// D(bi, bj) -= P_bi P_bj^T for one or two 16x16 blocks of the diagonal tile (K = 64)
__device__ __forceinline__ void syrk16(double* Ds, const double* Ps, int bi, int bj, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  double pi[16], pj[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) { pi[s] = Ps[(16 * bi + li) * GLD + 4 * s + lk]; pj[s] = Ps[(16 * bj + li) * GLD + 4 * s + lk]; }
  d4 acc = load_d16(Ds + 16 * bi * GLD + 16 * bj, GLD, lane);
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[s], pj[s], acc, 0, 0, 0);
  store_d16(Ds + 16 * bi * GLD + 16 * bj, GLD, acc, lane);
}
__device__ __forceinline__ void syrk16_own(double* Ds, const double* Ps, int w, int lane) {  // (w, 0) and (w, w)
  const int li = lane & 15, lk = lane >> 4;
  double pw[16], p0[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) { pw[s] = Ps[(16 * w + li) * GLD + 4 * s + lk]; p0[s] = Ps[li * GLD + 4 * s + lk]; }
  d4 a0 = load_d16(Ds + 16 * w * GLD, GLD, lane), aw = load_d16(Ds + 16 * w * GLD + 16 * w, GLD, lane);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pw[s], p0[s], a0, 0, 0, 0);
    aw = __builtin_amdgcn_mfma_f64_16x16x4f64(-pw[s], pw[s], aw, 0, 0, 0);
  }
  store_d16(Ds + 16 * w * GLD, GLD, a0, lane);
  store_d16(Ds + 16 * w * GLD + 16 * w, GLD, aw, lane);
}
// one 16x16 block of a tile from the accumulator layout straight to memory (system scope, like store_tile_coh)
typedef int i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_block_coh(double* G, size_t ld, int rb, int cb, d4 v, int lane) {
  const __amdgpu_buffer_rsrc_t r = tile_rsrc(G);
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double x = v[q];
    i2v w;
    w[0] = __double2loint(x); w[1] = __double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)(((size_t)(16 * rb + lk + 4 * q) * ld + 16 * cb + li) * 8), 0, kCoherent);
  }
}
// flags between the waves of one work-group, in LDS (the LDS unit executes a wave's accesses in order)
__device__ __forceinline__ void lds_flag_set(volatile int* f, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  *f = v;
}
__device__ __forceinline__ void lds_flag_wait(const volatile int* f, int v) {
  while (*f != v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void publish(unsigned* f, unsigned epoch) {
  __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls; false = the launch is being abandoned
__device__ __forceinline__ bool wait_flag(const unsigned* f, unsigned epoch, unsigned* abort_flag) {
  for (int spin = 0;; ++spin) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    if ((spin & 31) == 31) {
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return false;
      if (spin > kSpinLimit) { publish(abort_flag, epoch); return false; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
// the wave's 2x2 MFMA tiles (D layout) of a 64x64 LDS tile
__device__ __forceinline__ void quadrant_from_lds(const double* S, int wr, int wc, int lane, d4 acc[2][2]) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = S[(wr + 16 * m + lk + 4 * r) * GLD + wc + 16 * n + li];
}
__device__ __forceinline__ void quadrant_sub(d4 acc[2][2], const d4 p[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] -= p[m][n];
}
}  // namespace

struct CholPersistArgs {
  const double* M; double* L; double* inv; double* pre;  // M, L: tile stores (slot = tile_id)
  int nb;
  const CholTask* tasks; const int* wg_begin; const int* upd; const int* tile_id; const int* chain_info;
  unsigned* lflag; unsigned* dflag; unsigned* pflag; unsigned* abort_flag;
  unsigned epoch;
  double* fail;
  int drop_wg;                // test hook (MAVBA_CHOL_TEST_DROP_WG): this work-group does nothing, as if it were never resident
  unsigned long long* trace;  // MAVBA_CHOL_TRACE: 100 MHz wall-clock stamps, [8 per chain column | 4 per task], else null
};

__global__ void __launch_bounds__(256) k_chol_persist(CholPersistArgs A) {
  __shared__ __attribute__((aligned(16))) double As[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Bs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Cs[NB * GLD];
  __shared__ __attribute__((aligned(16))) double Ds[NB * GLD];  // the chain's diagonal tile (As still holds the sub-diagonal one)
  __shared__ int s_ok, s_ok2;
  __shared__ int s_rows[4];  // per wave: the chain column (+ 1) whose row block of the panel tile is complete in Cs
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  const int nb = A.nb;
  if (tid < 4) s_rows[tid] = 0;
  __syncthreads();
  constexpr size_t ld = NB;  // (pitch inside a tile of the store)
  const unsigned ep = A.epoch;
  // waits: lane 0 polls, everyone learns the outcome behind a barrier (which also orders the LDS reuse)
  auto wait2 = [&](const unsigned* f0, const unsigned* f1) -> bool {
    if (tid == 0) {
      bool ok = f0 ? wait_flag(f0, ep, A.abort_flag) : true;
      if (ok && f1) ok = wait_flag(f1, ep, A.abort_flag);
      s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    __syncthreads();  // s_ok may be rewritten by the next wait
    return ok;
  };
  if ((int)blockIdx.x == A.drop_wg) return;
  bool alive = true;
  auto stamp = [&](size_t slot) { if (A.trace && tid == 0) A.trace[slot] = wall_clock64(); };
  // (slots 3 / 4 of a chain column: the SHADER clock before and after the tile factorisation - against the 100 MHz stamps 2 / 6
  // they give the clock the launch really runs at)
  auto stamp_clk = [&](size_t slot) { if (A.trace && tid == 0) A.trace[slot] = (unsigned long long)clock64(); };
  for (int ti = A.wg_begin[blockIdx.x]; alive && ti < A.wg_begin[blockIdx.x + 1]; ++ti) {
    const CholTask T = A.tasks[ti];
    const size_t tslot = (size_t)8 * nb + (size_t)4 * ti;
    if (T.kind != CHOL_TASK_CHAIN) {
      stamp(tslot);
      // ---- owner-computes tile task: tile (i, j), updates upd[ub, ue) ----
      const int i = T.i, j = T.j;
      d4 acc[2][2], p[2][2];
      {
        const double* C = A.M + ((size_t)A.tile_id[(size_t)i * nb + j] << 12);  // written before this launch: plain loads
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = C[(size_t)(wr + 16 * m + lk + 4 * r) * ld + wc + 16 * n + li];
      }
      // (Round 6, measured and dropped - profiles/r06_ab_update_prefetch.txt: requesting update u + 1's two tiles into registers
      // before update u's product - a look at their flags without waiting, loads in flight under the 64 matrix instructions -
      // took C5's launch from 1.97 to 1.87 ms and made C3's and C2's 2 % SLOWER (0.264 -> 0.270, 0.101 -> 0.103 ms: one more
      // round trip before a barrier on lists whose inputs are rarely there early); with the look issued an iteration ahead
      // it lost at C5 too. C5's helpers are bound by the 4.7 GB of system-scope tile reads, not by their latency.)
      for (int u = T.ub; u < T.ue; ++u) {
        const int k = A.upd[u];
        const int s_ik = A.tile_id[(size_t)i * nb + k], s_jk = i != j ? A.tile_id[(size_t)j * nb + k] : 0;  // flag index = slot
        const unsigned* f0 = A.lflag + s_ik;
        const unsigned* f1 = i != j ? A.lflag + s_jk : nullptr;
        if (!wait2(f0, f1)) { alive = false; break; }
        load_tile_coh(A.L + ((size_t)s_ik << 12), ld, As, tid);
        if (i != j) load_tile_coh(A.L + ((size_t)s_jk << 12), ld, Bs, tid);
        __syncthreads();
        mfma_quadrant_nt(As, i != j ? Bs : As, wr, wc, lane, p);
        quadrant_sub(acc, p);
      }
      if (!alive) break;
      stamp(tslot + 1);
      if (T.kind == CHOL_TASK_TILE) {
        // panel solve against L_jj^-1, publish L_ij
        if (!wait2(A.dflag + j, nullptr)) { alive = false; break; }
        stamp(tslot + 2);
        quadrant_to_lds(As, wr, wc, lane, acc);
        load_tile_coh(A.inv + (size_t)j * NB * NB, NB, Bs, tid);
        __syncthreads();
        trsm_lower_tri(As, Bs, Cs, wv, lane);
        __syncthreads();
        const int s_ij = A.tile_id[(size_t)i * nb + j];
        store_tile_coh(A.L + ((size_t)s_ij << 12), ld, Cs, tid);
        drain_stores();
        __syncthreads();
        if (tid == 0) publish(A.lflag + s_ij, ep);
        stamp(tslot + 3);
      } else {
        // PRE: the chain's tile with every update but the chain's own; slot 2 j (diagonal) / 2 j + 1 (sub-diagonal)
        const int slot = T.kind == CHOL_TASK_PRE_DIAG ? 2 * j : 2 * i + 1;
        __syncthreads();  // the last product's LDS reads are done
        quadrant_to_lds(As, wr, wc, lane, acc);
        __syncthreads();
        store_tile_coh(A.pre + (size_t)slot * NB * NB, NB, As, tid);
        drain_stores();
        __syncthreads();
        if (tid == 0) publish(A.pflag + slot, ep);
        stamp(tslot + 3);
      }
      continue;
    }
    // ---- chain task: the diagonal of one tree node, tile columns [T.i, T.j) ----
    // LDS roles: Bs = L_jj^-1 of the column just factorised, Cs = the panel tile P_{j, j-1}, As = sub-diagonal tile,
    // then the diagonal tile. The two tiles of the NEXT column (left by the helpers' PRE tasks, or untouched in M)
    // are requested before the tile factorisation starts whenever their flags are already up - the usual case - and
    // travel while the matrix cores factorise; otherwise the chain waits for them afterwards.
    TileRegs Rsub, Rdiag;
    bool have_next = false;  // Rsub / Rdiag hold column j's tiles
    // slots (tile store) of the column's diagonal and sub-diagonal tile: looked up a column ahead, so that the dependent
    // scalar load is long done when the chain needs the address (tile store, round 6: + 0.4 us per column without this)
    int s_diag = A.tile_id[(size_t)T.i * nb + T.i], s_sub = 0, s_ndiag = 0, s_nsub = 0;
    for (int j = T.i; j < T.j; ++j, s_diag = s_ndiag, s_sub = s_nsub) {
      const int info = A.chain_info[j];
      const bool sub = j > T.i;
      if (j + 1 < T.j) { s_ndiag = A.tile_id[(size_t)(j + 1) * nb + (j + 1)]; s_nsub = A.tile_id[(size_t)(j + 1) * nb + j]; }
      stamp((size_t)8 * j);
      if (!have_next) {
        const unsigned* f0 = (sub && (info & 2)) ? A.pflag + 2 * j + 1 : nullptr;
        const unsigned* f1 = (info & 1) ? A.pflag + 2 * j : nullptr;
        if ((f0 || f1) && !wait2(f0, f1)) { alive = false; break; }
        if (sub) {
          if (info & 2) load_tile_regs(A.pre + (size_t)(2 * j + 1) * NB * NB, NB, Rsub, tid, true);
          else load_tile_regs(A.M + ((size_t)s_sub << 12), ld, Rsub, tid, false);
        }
        if (info & 1) load_tile_regs(A.pre + (size_t)(2 * j) * NB * NB, NB, Rdiag, tid, true);
        else load_tile_regs(A.M + ((size_t)s_diag << 12), ld, Rdiag, tid, false);
      }
      have_next = false;
      stamp((size_t)8 * j + 1);
      // Column j with a sub-diagonal tile: P = A_{j,j-1} L_{j-1,j-1}^-T and D_j -= P P^T are cut so that only what the FIRST
      // 16 pivots need precedes them: all four waves solve the first 16 rows of P (one column block each, <= 16 matrix
      // instructions), wave 0 updates block (0, 0) of D_j and starts factorising; the other 48 rows of P (40 instructions
      // per wave) and the other nine blocks of D_j (48 per wave) are done by waves 1..3 in phase 0 of the tile
      // factorisation, where they used to idle. The panel tile goes to memory block by block, straight from the
      // accumulators; it is drained after phase 0 and published after phase 1.
      const bool more = j + 1 < T.j;
      const int seq = j + 1;
      // Round 5. A node's FIRST column has no panel tile: it is the plain systolic tile factorisation (7.4 us against the
      // look-ahead variant's 10.3 at the persistent launch's clock). For the columns WITH a panel tile the fused form below
      // (tile_potrf_inv_sys + ChainPanel: panel solve and diagonal update in the prologue, every wave on its own program)
      // was built, tested and measured - 10.3 us per column against 9.2 + 1.5 for the round-3 arrangement that hides the
      // panel work inside the look-ahead variant's longer phases: wave 0 cannot start on the pivots before it has solved
      // its 16 rows of P and downdated block (0, 0) (56 matrix instructions, 4 100 ticks), and spreading that over the
      // waves makes wave 1 late for ITS pivots (11.1 us). No gain: those columns keep the round-3 code; the fused form
      // (ChainPanel) is in scripts/_dbg/pruned_r06.patch.
      if (!sub) {
        (void)seq;
        tile_regs_to_lds(Rdiag, Ds, tid);
        __syncthreads();
        stamp((size_t)8 * j + 2);
        stamp_clk((size_t)8 * j + 3); stamp((size_t)8 * j + 5);
        bool stalled = false;
        const bool ok = tile_potrf_inv_sys(Ds, Bs, tid, NoMark(), NoPanel(), &stalled);
        // (a wave of the tile that never got its record - ~64 k spins - is a stall of this launch, not a bad pivot: reported like
        // the launch's own time-outs, so the solve is repeated on the launch-per-panel schedule instead of being taken for "not
        // positive definite" - ADVICE r5)
        if (tid == 0 && !ok) atomicAdd(A.fail, stalled ? 1e30 : 1.0);
        stamp_clk((size_t)8 * j + 4);
        stamp((size_t)8 * j + 6);
        store_tile_coh(A.inv + (size_t)j * NB * NB, NB, Bs, tid);
        drain_stores();
        __syncthreads();
        if (tid == 0) publish(A.dflag + j, ep);
        stamp((size_t)8 * j + 7);
        continue;
      }
      if (sub) {
        tile_regs_to_lds(Rsub, As, tid);
        tile_regs_to_lds(Rdiag, Ds, tid);
        __syncthreads();
        double* Pg = A.L + ((size_t)s_sub << 12);
        // rows 0..31 of P now, 20 matrix instructions per wave: blocks (0,0)+(1,3) | (0,1)+(1,2) | (0,2)+(1,1) | (0,3)+(1,0)
        d4 p0, p1;
        switch (wv) {
          case 0: p0 = trsm_block16<0>(As, Bs, 0, lane); p1 = trsm_block16<3>(As, Bs, 1, lane); break;
          case 1: p0 = trsm_block16<1>(As, Bs, 0, lane); p1 = trsm_block16<2>(As, Bs, 1, lane); break;
          case 2: p0 = trsm_block16<2>(As, Bs, 0, lane); p1 = trsm_block16<1>(As, Bs, 1, lane); break;
          default: p0 = trsm_block16<3>(As, Bs, 0, lane); p1 = trsm_block16<0>(As, Bs, 1, lane); break;
        }
        store_d16(Cs + 16 * wv, GLD, p0, lane);
        store_d16(Cs + 16 * GLD + 16 * (3 - wv), GLD, p1, lane);
        store_block_coh(Pg, ld, 0, wv, p0, lane);
        store_block_coh(Pg, ld, 1, 3 - wv, p1, lane);
        __syncthreads();
      } else {
        tile_regs_to_lds(Rdiag, Ds, tid);
        __syncthreads();
      }
      stamp((size_t)8 * j + 2);
      stamp_clk((size_t)8 * j + 3);
      stamp((size_t)8 * j + 5);
      auto next_ready = [&]() {
        const int ni = A.chain_info[j + 1];
        int ready = 1;
        if ((ni & 2) && __hip_atomic_load(A.pflag + 2 * (j + 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ep) ready = 0;
        if ((ni & 1) && __hip_atomic_load(A.pflag + 2 * (j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ep) ready = 0;
        return ready;
      };
      auto request_next = [&]() {
        const int ni = A.chain_info[j + 1];
        if (ni & 2) load_tile_regs(A.pre + (size_t)(2 * (j + 1) + 1) * NB * NB, NB, Rsub, tid, true);
        else load_tile_regs(A.M + ((size_t)s_nsub << 12), ld, Rsub, tid, false);
        if (ni & 1) load_tile_regs(A.pre + (size_t)(2 * (j + 1)) * NB * NB, NB, Rdiag, tid, true);
        else load_tile_regs(A.M + ((size_t)s_ndiag << 12), ld, Rdiag, tid, false);
        have_next = true;
      };
      // hook(ph): every thread, right after the barrier that ends phase ph of the factorisation
      //   0: the panel tile's stores are drained; wave 3 polls the flags of the next column's two tiles
      //   1: the panel tile is published; the next column's loads go out if the helpers were done, else ...
      //   2: ... wave 3 polls again and   3: the loads go out now (they land during the last phase)
      // (measured and dropped, A/B on one box: publishing after phase 0 behind one more barrier - 0.4 us per column for
      // nothing; a third prefetch attempt, polled after phase 1 - 3 % slower; moving update blocks (2,2) into phase 0 and
      // (3,2), (3,3) into phase 2 - no difference)
      auto in_factor = [&](int ph) {
        if (ph == 0) {
          if (sub) drain_stores();
          if (more && tid == 192) s_ok = next_ready();
        } else if (ph == 1) {
          if (sub && tid == 0) publish(A.lflag + s_sub, ep);
          if (more && s_ok) request_next();
        } else if (ph == 2) {
          if (more && !have_next && tid == 192) s_ok2 = next_ready();
        } else if (ph == 3) {
          if (more && !have_next && s_ok2) request_next();
        }
      };
      auto phase0 = [&](int stage) {
        if (!sub) return;
        if (stage == 0) {
          syrk16(Ds, Cs, 0, 0, lane);  // wave 0: block (0, 0) of the diagonal update (row block 0 of P is complete)
          wave_lds_sync();
        } else if (stage == 1) {
          // the first block of the new inverse overwrites Bs: the other waves must be done reading the old one
          lds_flag_wait(&s_rows[1], seq); lds_flag_wait(&s_rows[2], seq); lds_flag_wait(&s_rows[3], seq);
        } else if (stage == 2) {
          if (wv == 1) {
            syrk16_own(Ds, Cs, 1, lane);  // (1, 0), (1, 1): rows 0..31 of P are complete
            if (lane == 0) lds_flag_set(&s_rows[1], seq);  // (this wave does not read Bs in phase 0)
          } else {
            double* Pg = A.L + ((size_t)s_sub << 12);
            d4 c[4];
            trsm_row16(As, Bs, wv, lane, c);
#pragma unroll
            for (int n = 0; n < 4; ++n) { store_d16(Cs + 16 * wv * GLD + 16 * n, GLD, c[n], lane); store_block_coh(Pg, ld, wv, n, c[n], lane); }
            if (lane == 0) lds_flag_set(&s_rows[wv], seq);  // (Bs is no longer read by this wave)
            wave_lds_sync();
            syrk16(Ds, Cs, wv, 0, lane);  // (wv, 0): what phase 1 reads
          }
        } else {
          // start of phase 1 (all of P is complete): the blocks this wave itself updates further in phases 1..3
          if (wv == 1) { syrk16(Ds, Cs, 2, 1, lane); syrk16(Ds, Cs, 2, 2, lane); }
          else if (wv == 2) { syrk16(Ds, Cs, 3, 1, lane); syrk16(Ds, Cs, 3, 3, lane); }
          else syrk16(Ds, Cs, 3, 2, lane);
          wave_lds_sync();
        }
      };
      const bool ok = tile_potrf_inv_la(Ds, Bs, tid, in_factor, NoMark(), phase0);
      if (tid == 0 && !ok) atomicAdd(A.fail, 1.0);
      stamp_clk((size_t)8 * j + 4);
      stamp((size_t)8 * j + 6);
      store_tile_coh(A.inv + (size_t)j * NB * NB, NB, Bs, tid);
      drain_stores();
      __syncthreads();
      if (tid == 0) publish(A.dflag + j, ep);
      stamp((size_t)8 * j + 7);
    }
  }
  if (!alive && tid == 0) atomicAdd(A.fail, 1e30);
}

void CholStructure::release() {
  if (d_trace) {
    // MAVBA_CHOL_TRACE=<file>: the stamps of the LAST solve, with the schedule they belong to
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)8 * nb + (size_t)4 * h_tasks.size());
    if (hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = std::fopen(std::getenv("MAVBA_CHOL_TRACE"), "w")) {
        std::fprintf(f, "# nb %d grid %d chain_wgs %d tiles %lld updates %lld\n", nb, persist_grid, persist_chain_wgs, persist_tiles, persist_updates);
        for (int j = 0; j < nb; ++j) {
          std::fprintf(f, "C %d %d", j, seg_of_tile[j]);
          for (int q = 0; q < 8; ++q) std::fprintf(f, " %llu", h[(size_t)8 * j + q]);
          std::fprintf(f, "\n");
        }
        for (size_t t = 0; t < h_tasks.size(); ++t) {
          int wg = 0;
          while (wg + 1 < (int)h_wg_begin.size() && h_wg_begin[wg + 1] <= (int)t) ++wg;
          std::fprintf(f, "T %d %d %d %d %d", wg, h_tasks[t].kind, h_tasks[t].i, h_tasks[t].j, h_tasks[t].ue - h_tasks[t].ub);
          for (int q = 0; q < 4; ++q) std::fprintf(f, " %llu", h[(size_t)8 * nb + 4 * t + q]);
          std::fprintf(f, "\n");
        }
        std::fclose(f);
      }
    device_free(d_trace);
    d_trace = nullptr;
  }
  if (d_ints) device_free(d_ints);
  if (d_fronts) device_free(d_fronts);
  if (d_shadow) device_free(d_shadow);
  if (d_merges) device_free(d_merges);
  if (d_tasks) device_free(d_tasks);
  if (d_pints) device_free(d_pints);
  if (d_pflags) device_free(d_pflags);
  if (d_pre) device_free(d_pre);
  if (d_tile_slot) device_free(d_tile_slot);
  d_tile_slot = nullptr;
  d_ints = nullptr; d_fronts = nullptr; d_shadow = nullptr; d_merges = nullptr;
  d_tasks = nullptr; d_pints = nullptr; d_pflags = nullptr; d_pre = nullptr;
  persist_ok = false;
}
CholStructure::~CholStructure() { release(); }

hipError_t CholStructure::build_dense(int nb_) {
  std::vector<std::pair<int, int>> pairs;
  for (int i = 0; i < nb_; ++i) pairs.emplace_back(i, 0);  // every row starts at tile 0
  return build(nb_, pairs, {}, 0);
}

hipError_t CholStructure::build(int nb_, const std::vector<std::pair<int, int>>& tile_pairs,
                                const std::vector<CholNode>& tree_in, hipStream_t st) {
  release();
  nb = nb_;
  const int never = nb + 1;
  nodes = tree_in;
  // MAVBA_CHOL_DUMP=<file>: the structure as handed over (tile columns, tree, non-zero lower tiles) - the input of
  // mavba_debug_chol_schedule, for working on the schedule of a real problem without a device
  if (const char* dump = host_only ? nullptr : std::getenv("MAVBA_CHOL_DUMP")) {
    if (FILE* f = std::fopen(dump, "w")) {
      std::fprintf(f, "%d %zu %zu\n", nb, tree_in.size(), tile_pairs.size());
      for (const CholNode& n : tree_in) std::fprintf(f, "%d %d %d\n", n.begin, n.end, n.parent);
      for (const auto& pr : tile_pairs) std::fprintf(f, "%d %d\n", pr.first, pr.second);
      std::fclose(f);
    }
  }
  // valid tree: contiguous ascending cover of [0, nb), parents after children, exactly one root (the last node)
  auto tree_ok = [&]() {
    if (nodes.empty() || nb > kMaxBacksolveGroups) return false;  // (the per-tile backward fallback is single-segment)
    int at = 0;
    for (size_t n = 0; n < nodes.size(); ++n) {
      if (nodes[n].begin != at || nodes[n].end <= nodes[n].begin) return false;
      at = nodes[n].end;
      const bool last = n + 1 == nodes.size();
      if (last ? nodes[n].parent != -1 : (nodes[n].parent <= (int)n || nodes[n].parent >= (int)nodes.size())) return false;
    }
    return at == nb;
  };
  std::vector<int> height;
  auto is_ancestor = [&](int a, int d) {  // a == d or a above d
    while (d != -1 && d != a) d = nodes[d].parent;
    return d == a;
  };
  auto assign_segments = [&]() {
    nseg = (int)nodes.size();
    seg_of_tile.assign(nb, 0);
    for (int n = 0; n < nseg; ++n)
      for (int t = nodes[n].begin; t < nodes[n].end; ++t) seg_of_tile[t] = n;
    seg_first.assign((size_t)nb * nseg, never);
    for (int i = 0; i < nb; ++i) seg_first[(size_t)i * nseg + seg_of_tile[i]] = i;
    for (const auto& pr : tile_pairs) {
      const int tr = pr.first, tc = pr.second;
      const int q = seg_of_tile[tc];
      if (!is_ancestor(seg_of_tile[tr], q)) return false;  // coupling across two branches: not a valid dissection
      int& f = seg_first[(size_t)tr * nseg + q];
      f = std::min(f, tc);
    }
    return true;
  };
  if (!tree_ok() || !assign_segments()) {
    nodes.assign(1, CholNode{0, nb, -1});
    assign_segments();
  }
  height.assign(nseg, 0);
  std::vector<char> is_leaf(nseg, 1);
  for (int n = 0; n < nseg; ++n)
    if (nodes[n].parent >= 0) { is_leaf[nodes[n].parent] = 0; height[nodes[n].parent] = std::max(height[nodes[n].parent], height[n] + 1); }
  // Separators fill in: dense inside, and an ancestor row that couples to any descendant of a separator couples
  // to the whole separator (children precede parents, so their entries are final when the parent is visited).
  if (nseg > 1)
    for (int n = 0; n < nseg; ++n) {
      if (is_leaf[n]) continue;
      for (int i = nodes[n].begin; i < nodes[n].end; ++i) seg_first[(size_t)i * nseg + n] = nodes[n].begin;
      for (int i = nodes[n].end; i < nb; ++i) {
        bool coupled = seg_first[(size_t)i * nseg + n] != never;
        for (int d = 0; d < n && !coupled; ++d)
          coupled = is_ancestor(n, d) && seg_first[(size_t)i * nseg + d] != never;
        if (coupled) seg_first[(size_t)i * nseg + n] = nodes[n].begin;
      }
    }

  // shadow blocks: every non-root node except the first of its level writes its ancestor updates to its own
  // block, origin = its parent's first tile, (nb - origin + 1) x (nb - origin) tiles
  const int H = *std::max_element(height.begin(), height.end());
  std::vector<long long> sh_off(nseg, -1);
  std::vector<int> sh_begin(nseg, 0);
  shadow_doubles = 0;
  for (int hgt = 0; hgt < H; ++hgt) {
    bool first = true;
    for (int n = 0; n < nseg; ++n) {
      if (height[n] != hgt || nodes[n].parent < 0) continue;
      sh_begin[n] = nodes[nodes[n].parent].begin;
      if (first) { first = false; continue; }
      const size_t ns = (size_t)(nb - sh_begin[n]);
      sh_off[n] = (long long)shadow_doubles;
      shadow_doubles += (ns + 1) * ns * 4096;
    }
  }

  std::vector<int> rows;
  fronts.clear(); steps.clear(); init_tiles.clear(); merges.clear();
  envelope_tiles = 0; factor_flops = 0.0; chain_steps = 0; num_fronts_max = 1;
  const double t3 = 64.0 * 64.0 * 64.0;
  auto add_front = [&](std::vector<CholFront>& cur, int k, int n) {
    CholFront F;
    F.k = k; F.act_off = (int)rows.size(); F.factor_next = (k + 1 < nodes[n].end) ? 1 : 0;
    F.sh_begin = sh_begin[n]; F.sh_off = sh_off[n];
    // Row k + 1 leads the list whenever the front goes on (its diagonal tile is factorised by the owner of
    // tile (k+1, k+1)); then the rows of the same node whose envelope reaches panel k, then the rows of its
    // ancestors that couple to it (rows of other branches never do).
    if (F.factor_next) rows.push_back(k + 1);
    for (int i = k + 2; i < nodes[n].end; ++i) if (seg_first[(size_t)i * nseg + n] <= k) rows.push_back(i);
    for (int i = nodes[n].end; i < nb; ++i) if (seg_first[(size_t)i * nseg + n] <= k) rows.push_back(i);
    F.na = (int)rows.size() - F.act_off;
    envelope_tiles += 1 + F.na;
    const double na = F.na;
    // tile factor + inverse (~2/3 t3), panel solves (na + rhs) * 2 t3, trailing update incl. rhs row
    factor_flops += (2.0 / 3.0) * t3 + (na + 1.0) * 2.0 * t3 + (na * (na + 1.0) / 2.0 + na) * 2.0 * t3;
    cur.push_back(F);
  };
  auto close_step = [&](std::vector<CholFront>& cur) {
    std::stable_sort(cur.begin(), cur.end(), [](const CholFront& a, const CholFront& b) { return (a.na > 0) > (b.na > 0); });
    CholStep S{};
    S.kind = 0; S.front_off = (int)fronts.size();
    for (const CholFront& F : cur) { if (F.na > 0) ++S.nf; else ++S.nf0; S.max_na = std::max(S.max_na, F.na); S.tasks += F.na * (F.na + 1) / 2 + F.na; }
    fronts.insert(fronts.end(), cur.begin(), cur.end());
    steps.push_back(S);
    cur.clear();
  };
  std::vector<CholFront> cur;
  for (int n = 0; n < nseg; ++n) if (height[n] == 0) init_tiles.push_back(nodes[n].begin);
  num_leaf_init = (int)init_tiles.size();
  for (int hgt = 0; hgt <= H; ++hgt) {
    int lead = 0, width = 0;
    for (int n = 0; n < nseg; ++n) if (height[n] == hgt) { lead = std::max(lead, nodes[n].end - nodes[n].begin); ++width; }
    num_fronts_max = std::max(num_fronts_max, width);
    for (int s = 0; s < lead; ++s) {
      for (int n = 0; n < nseg; ++n)
        if (height[n] == hgt && nodes[n].begin + s < nodes[n].end) add_front(cur, nodes[n].begin + s, n);
      close_step(cur);
    }
    chain_steps += lead;
    if (hgt == H) break;
    // end of the level: merge its shadows, then start the next level's nodes
    CholStep Mg{};
    Mg.kind = 1; Mg.front_off = (int)merges.size(); Mg.merge_begin = nb; Mg.init_off = (int)init_tiles.size();
    for (int n = 0; n < nseg; ++n)
      if (height[n] == hgt && sh_off[n] >= 0) { merges.push_back(CholMerge{sh_begin[n], sh_off[n]}); ++Mg.nf; Mg.merge_begin = std::min(Mg.merge_begin, sh_begin[n]); }
    for (int n = 0; n < nseg; ++n) if (height[n] == hgt + 1) { init_tiles.push_back(nodes[n].begin); ++Mg.nf0; }
    steps.push_back(Mg);
  }

  // rows that couple to every column, schedule step of every column
  std::vector<std::vector<int>> col_rows(nb);
  std::vector<int> col_step(nb, 0);
  {
    int ordinal = 0;
    for (const CholStep& S : steps) {
      if (S.kind != 0) continue;
      for (int f = S.front_off; f < S.front_off + S.nf + S.nf0; ++f) {
        const CholFront& F = fronts[f];
        col_rows[F.k].assign(rows.begin() + F.act_off, rows.begin() + F.act_off + F.na);
        col_step[F.k] = ordinal;
      }
      ++ordinal;
    }
  }
  // Tile store (round 6): the matrix and the factor keep the envelope's tiles only - per column the diagonal tile, the coupled
  // rows, the right-hand-side tile (tile row nb). The slot is also the tile's flag index in the persistent launch.
  tile_slot.assign((size_t)(nb + 1) * nb, -1);
  num_tiles = 0;
  for (int k = 0; k < nb; ++k) {
    tile_slot[(size_t)k * nb + k] = (int)num_tiles++;
    for (int i : col_rows[k]) tile_slot[(size_t)i * nb + k] = (int)num_tiles++;
    tile_slot[(size_t)nb * nb + k] = (int)num_tiles++;
  }
  // (the launch-per-panel schedule updates tile (i, j) for every pair of a front's rows: all of them must be in the store)
  for (const CholFront& F : fronts)
    for (int a = 0; a < F.na; ++a)
      for (int b2 = 0; b2 <= a; ++b2) {
        const int i = rows[(size_t)F.act_off + a], j = rows[(size_t)F.act_off + b2];
        if (tile_slot[(size_t)std::max(i, j) * nb + std::min(i, j)] < 0) {
          std::fprintf(stderr, "mavba: factorisation envelope is not closed under its own updates (tile %d, %d of front %d)\n", i, j, F.k);
          release();
          return hipErrorInvalidValue;
        }
      }
  // device copies: rows | seg_of_tile | seg_first | init_tiles | flags
  std::vector<int> pack(rows);
  const size_t o_seg = pack.size();
  pack.insert(pack.end(), seg_of_tile.begin(), seg_of_tile.end());
  const size_t o_first = pack.size();
  pack.insert(pack.end(), seg_first.begin(), seg_first.end());
  const size_t o_init = pack.size();
  pack.insert(pack.end(), init_tiles.begin(), init_tiles.end());
  const size_t o_flags = pack.size();
  pack.resize(pack.size() + nb, 0);
  hipError_t e = hipSuccess;
  if (!host_only) {
  e = device_alloc(reinterpret_cast<void**>(&d_ints), pack.size() * sizeof(int));
  if (e == hipSuccess) e = copy_h2d_staged(d_ints, pack.data(), pack.size() * sizeof(int), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_fronts), std::max<size_t>(fronts.size(), 1) * sizeof(CholFront));
  if (e == hipSuccess && !fronts.empty())
    e = copy_h2d_staged(d_fronts, fronts.data(), fronts.size() * sizeof(CholFront), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_merges), std::max<size_t>(merges.size(), 1) * sizeof(CholMerge));
  if (e == hipSuccess && !merges.empty())
    e = copy_h2d_staged(d_merges, merges.data(), merges.size() * sizeof(CholMerge), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_tile_slot), tile_slot.size() * sizeof(int));
  if (e == hipSuccess) e = copy_h2d_staged(d_tile_slot, tile_slot.data(), tile_slot.size() * sizeof(int), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // the staging vectors go out of scope
  release_staged(st);
  if (e != hipSuccess) { release(); return e; }
  d_rows = d_ints;
  d_seg_of_tile = d_ints + o_seg;
  d_seg_first = d_ints + o_first;
  d_init = d_ints + o_init;
  d_flags = reinterpret_cast<unsigned*>(d_ints + o_flags);
  }  // (!host_only)
  // persistent schedule
  e = build_persistent(col_rows, col_step, height, st);
  if (e != hipSuccess) { release(); return e; }
  return hipSuccess;
}

namespace {
int device_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chol_persist, 256, 0) != hipSuccess || per_cu < 1) return 0;
    cus[dev] = prop.multiProcessorCount;
  }
  return cus[dev];
}
}  // namespace

hipError_t CholStructure::build_persistent(const std::vector<std::vector<int>>& col_rows, const std::vector<int>& col_step,
                                           const std::vector<int>& height, hipStream_t st) {
  persist_ok = false;
  const bool sched_debug = std::getenv("MAVBA_CHOL_SCHED_DEBUG") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!sched_debug) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "mavba:   schedule builder: %-28s %8.3f ms\n", what, 1e3 * std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  // MAVBA_CHOL_PERSIST: 0 = never, 1 (default) = when it pays, 2 = whenever a schedule exists
  static const int mode = [] { const char* e = std::getenv("MAVBA_CHOL_PERSIST"); return e ? std::atoi(e) : 1; }();
  const bool enabled = mode != 0;
  static const int grid_cap = [] { const char* e = std::getenv("MAVBA_CHOL_PERSIST_GRID"); return e ? std::atoi(e) : 0; }();
  static const int max_nb = [] { const char* e = std::getenv("MAVBA_CHOL_PERSIST_MAX_NB"); return e ? std::atoi(e) : 1024; }();
  // (round 6: 512 -> 1024 tile columns - the old limit belonged to a dense tile table; a 10 000-image scene, 943 columns: 64.7 ->
  // 44.6 ms per LM iteration, the builder takes 55 ms of a 1.2 s set-up, profiles/r06_scale_probe.txt)
  if (!enabled || nb < 1 || nb > max_nb) return hipSuccess;  // (larger systems: the launch-per-panel schedule; the builder below is O(updates))
  int G = host_only ? host_only_cus : device_cu_count();
  if (grid_cap > 0) G = std::min(G, grid_cap);
  // ---- tiles with a 'published' flag: diagonal, coupled rows, right-hand-side row of every column ----
  // (= the slots of the tile store, CholStructure::build: a tile's flag and its place in memory share one index)
  const std::vector<int>& tile_id = tile_slot;
  const long long nt = num_tiles;
  // ---- the updates every tile receives, in a fixed order: by availability (schedule step of k), then k ----
  std::vector<std::vector<int>> upd_of((size_t)nt);
  std::vector<int> R;
  long long nupd = 0;
  for (int k = 0; k < nb; ++k) {
    R.assign(col_rows[k].begin(), col_rows[k].end());
    R.push_back(nb);
    for (size_t a = 0; a < R.size(); ++a)
      for (size_t b = 0; b <= a; ++b) {
        if (R[b] == nb) continue;  // (the right-hand-side row is no column)
        const int id = tile_id[(size_t)R[a] * nb + R[b]];
        if (id < 0) return hipSuccess;  // fill outside the envelope: keep the launch-per-panel schedule
        upd_of[id].push_back(k);
        ++nupd;
      }
  }
  for (auto& l : upd_of)
    std::sort(l.begin(), l.end(), [&](int a, int b) { return col_step[a] != col_step[b] ? col_step[a] < col_step[b] : a < b; });
  lap("update lists");
  // ---- a timing model of the launch (round 4) ----
  // The order of a tile's updates and the place of every task in the helpers' queues used to follow the launch-per-panel
  // step of the columns - the same number for the s-th columns of ALL nodes of a tree level. On C3's critical path that cost
  // ~20 us: the root's first tile applied the update of one child's last column (ready last) BEFORE the other child's last
  // two, and a tile with eight updates was queued behind another task and started 20 us late. Now the launch is simulated
  // on the host with measured costs (MAVBA_CHOL_TRACE: ~3.5 us per tile update of a helper, 3.1 us panel solve + publish,
  // ~12.2 us per chain column, 9.6 for a node's first): every update list is ordered by the time its inputs are ready, and
  // the helpers' queues are filled by list scheduling (below). Only the ORDER comes from the model, never correctness.
  // (round 5: a node's first column is the systolic tile factorisation - 7.4 us of factor + load and publish)
  constexpr double cU = 3.5, cS = 3.1, cP = 1.0, cCol = 12.2, cColFirst = 7.4, cSub = 4.0;
  std::vector<std::pair<int, int>> id_ij((size_t)nt, {0, 0});
  for (int k = 0; k < nb; ++k) {
    id_ij[tile_id[(size_t)k * nb + k]] = {k, k};
    for (int i : col_rows[k]) id_ij[tile_id[(size_t)i * nb + k]] = {i, k};
    id_ij[tile_id[(size_t)nb * nb + k]] = {nb, k};
  }
  std::vector<std::vector<int>> children(nseg);
  for (int n = 0; n < nseg; ++n) if (nodes[n].parent >= 0) children[nodes[n].parent].push_back(n);
  struct Times { std::vector<double> L, col_begin, col_fin, pre; };  // tile published | chain column | PRE slot ready
  // in_ready: when the two factor tiles that update k of tile (i, j) multiplies are both published
  auto in_ready = [&](const Times& T, int i, int j, int k) {
    double r = T.L[tile_id[(size_t)i * nb + k]];
    if (i != j) r = std::max(r, T.L[tile_id[(size_t)j * nb + k]]);
    return r;
  };
  // one task's run from `start` on: its updates in list order (each waits for its inputs); returns the end of the update phase
  auto run_updates_at = [&](const Times& T, int i, int j, const int* list, size_t count, double start) {
    double f = start;
    for (size_t q = 0; q < count; ++q) f = std::max(f, in_ready(T, i, j, list[q])) + cU;
    return f;
  };
  auto run_updates = [&](const Times& T, int i, int j, const std::vector<int>& list, size_t count, double start) {
    return run_updates_at(T, i, j, list.data(), count, start);
  };
  auto own_updates = [&](int j) {  // the diagonal tile's list without the chain's own (last) update
    const int n = seg_of_tile[j];
    const std::vector<int>& dl = upd_of[tile_id[(size_t)j * nb + j]];
    return j == nodes[n].begin ? dl.size() : (dl.empty() ? 0 : dl.size() - 1);
  };
  // chain column j: begins when its predecessor (or every child node) and its PRE tiles are there
  auto chain_begin = [&](Times& T, int j) {
    const int n = seg_of_tile[j];
    const bool first = j == nodes[n].begin;
    double start = 0.0;
    if (first) { for (int c : children[n]) start = std::max(start, T.col_fin[nodes[c].end - 1]); }
    else start = T.col_fin[j - 1];
    start = std::max(start, std::max(T.pre[2 * j], T.pre[2 * j + 1]));
    T.col_begin[j] = start;
    if (!first) T.L[tile_id[(size_t)j * nb + (j - 1)]] = start + cSub;  // (the chain solves and publishes its own panel tile early in the column)
  };
  auto chain_end = [&](Times& T, int j) { T.col_fin[j] = T.col_begin[j] + (j == nodes[seg_of_tile[j]].begin ? cColFirst : cCol); };
  auto chain_column = [&](Times& T, int j) { chain_begin(T, j); chain_end(T, j); };
  auto ideal_pass = [&](Times& T) {  // every helper task on a work-group of its own, started at time 0
    T.L.assign((size_t)nt, 0.0); T.col_begin.assign(nb, 0.0); T.col_fin.assign(nb, 0.0); T.pre.assign((size_t)2 * nb, 0.0);
    for (int j = 0; j < nb; ++j) {
      const int n = seg_of_tile[j];
      const bool first = j == nodes[n].begin, last = j + 1 == nodes[n].end;
      const size_t nd = own_updates(j);
      if (nd) T.pre[2 * j] = run_updates(T, j, j, upd_of[tile_id[(size_t)j * nb + j]], nd, 0.0) + cP;
      if (!first) {
        const std::vector<int>& sl = upd_of[tile_id[(size_t)j * nb + (j - 1)]];
        if (!sl.empty()) T.pre[2 * j + 1] = run_updates(T, j, j - 1, sl, sl.size(), 0.0) + cP;
      }
      chain_column(T, j);
      auto tile_task = [&](int i) {
        const int id = tile_id[(size_t)i * nb + j];
        T.L[id] = std::max(run_updates(T, i, j, upd_of[id], upd_of[id].size(), 0.0), T.col_fin[j]) + cS;
      };
      for (int i : col_rows[j]) if (!(i == j + 1 && !last)) tile_task(i);
      tile_task(nb);
    }
  };
  Times ideal;
  for (int round = 0; round < 2; ++round) {
    ideal_pass(ideal);
    for (long long id = 0; id < nt; ++id) {
      std::vector<int>& l = upd_of[id];
      const int i = id_ij[id].first, j = id_ij[id].second;
      const bool own_last = i == j && j != nodes[seg_of_tile[j]].begin && !l.empty() && l.back() == j - 1;  // (checked below)
      std::stable_sort(l.begin(), l.end() - (own_last ? 1 : 0), [&](int a, int b) {
        const double ra = in_ready(ideal, i, j, a), rb = in_ready(ideal, i, j, b);
        return ra != rb ? ra < rb : a < b;
      });
    }
  }
  ideal_pass(ideal);
  lap("ideal passes + list order");
  // ---- tasks ----
  struct Gen { CholTask t; long long key; int level; long long work; };
  std::vector<Gen> gen;
  std::vector<int> upd, chain_info(nb, 0);
  auto add_task = [&](int kind, int i, int j, const std::vector<int>& list, long long key) {
    CholTask t{kind, i, j, (int)upd.size(), (int)(upd.size() + list.size())};
    upd.insert(upd.end(), list.begin(), list.end());
    gen.push_back(Gen{t, key, height[seg_of_tile[j]], 2 + (long long)list.size()});
  };
  for (int j = 0; j < nb; ++j) {
    const int n = seg_of_tile[j];
    const bool first = j == nodes[n].begin, last = j + 1 == nodes[n].end;
    const long long sj = col_step[j];
    std::vector<int> dl = upd_of[tile_id[(size_t)j * nb + j]];
    if (!first) {
      if (dl.empty() || dl.back() != j - 1) return hipSuccess;  // (the chain's own update must come last)
      dl.pop_back();
    }
    if (!dl.empty()) { add_task(CHOL_TASK_PRE_DIAG, j, j, dl, (sj - 1) * 8 + 2); chain_info[j] |= 1; }
    if (!first) {
      const int id = tile_id[(size_t)j * nb + (j - 1)];
      if (id < 0) return hipSuccess;
      if (!upd_of[id].empty()) { add_task(CHOL_TASK_PRE_SUB, j, j - 1, upd_of[id], (sj - 1) * 8 + 2); chain_info[j] |= 2; }
    }
    for (int i : col_rows[j]) {
      if (i == j + 1 && !last) continue;  // the chain's sub-diagonal tile
      add_task(CHOL_TASK_TILE, i, j, upd_of[tile_id[(size_t)i * nb + j]], sj * 8 + (i <= j + 2 ? 4 : 5));
    }
    add_task(CHOL_TASK_TILE, nb, j, upd_of[tile_id[(size_t)nb * nb + j]], sj * 8 + 6);
  }
  // ---- work-groups: one chain per concurrent node, the helpers partitioned by tree level ----
  std::vector<int> chain_wg(nseg, -1);
  int nch = 0;
  {
    std::vector<int> first_child(nseg, -1);
    for (int n = 0; n < nseg; ++n)
      if (nodes[n].parent >= 0 && first_child[nodes[n].parent] < 0) first_child[nodes[n].parent] = n;
    for (int n = 0; n < nseg; ++n) chain_wg[n] = first_child[n] >= 0 ? chain_wg[first_child[n]] : nch++;
  }
  const int helpers = (int)std::min<long long>(G - nch, (long long)gen.size());
  if (helpers < 1) return hipSuccess;
  lap("tasks");
  // ---- list scheduling of the helper tasks on the simulated launch ----
  // Tasks and chain columns are visited in the order of their finishing times on unlimited helpers (`ideal`): every input of
  // a task finishes earlier there, so it has been placed - and given its time on the real number of helpers - before the task
  // itself. A task goes to the work-group that is free latest among those free by the time the task has to start in order to
  // finish on time (its updates back to back, ending with the last input's arrival; 10 us earlier for what the model does
  // not know) - or, if none is, to the one that is free first. A work-group runs its queue in this order and every wait is
  // for something that is earlier in it: no cycle of waits, whatever the real timing turns out to be.
  struct Event { double t; int chain; int idx; };  // idx: column (chain) or index into gen
  struct Pass { Times act; std::vector<std::pair<int, int>> placed; double forward = 0.0; bool ok = false; };  // placed: (helper, task) in visiting order
  auto forward_of = [&](const Times& T) {
    double f = 0.0;
    for (int j = 0; j < nb; ++j) f = std::max(f, T.col_fin[j]);
    for (size_t id = 0; id < T.L.size(); ++id) f = std::max(f, T.L[id]);  // (the last panel solves)
    return f;
  };
  struct OrderKeys { std::vector<double> task, begin, fin; };  // visiting times other than the estimate's own (null: those)
  // A visiting order = the events sorted by their times (the estimate's own, or `keys`), the end of every task's update phase in
  // the estimate when started at time 0, and the self-check of the order - all of it the same for every pool size that is
  // simulated on this order, so it is made once (it was two thirds of a pass).
  struct Order { std::vector<Event> events; std::vector<double> upd_end; bool ok = false; };
  auto make_order = [&](const Times& est, const OrderKeys* keys, Order& O) {
    std::vector<Event>& events = O.events;
    events.clear();
    events.reserve(gen.size() + 2 * (size_t)nb);
    O.upd_end.assign(gen.size(), 0.0);
    for (size_t g = 0; g < gen.size(); ++g) {
      const CholTask& t = gen[g].t;
      O.upd_end[g] = run_updates_at(est, t.i, t.j, upd.data() + t.ub, (size_t)(t.ue - t.ub), 0.0);
      const double fin = t.kind == CHOL_TASK_TILE ? est.L[tile_id[(size_t)t.i * nb + t.j]] : est.pre[t.kind == CHOL_TASK_PRE_DIAG ? 2 * t.j : 2 * t.i + 1];
      events.push_back(Event{keys ? keys->task[g] : fin, 0, (int)g});
    }
    // (a chain column is two events: its begin - from then on its own panel tile is on its way, tasks that multiply with it may
    // be visited before the column ends - and its end; at equal times: ends, then helper tasks, then begins)
    for (int j = 0; j < nb; ++j) {
      events.push_back(Event{keys ? keys->fin[j] : est.col_fin[j], 1, j});
      events.push_back(Event{keys ? keys->begin[j] : est.col_begin[j], 2, j});
    }
    std::stable_sort(events.begin(), events.end(), [&](const Event& x, const Event& y) {
      if (x.t != y.t) return x.t < y.t;
      const int rx = x.chain == 1 ? 0 : (x.chain == 0 ? 1 : 2), ry = y.chain == 1 ? 0 : (y.chain == 0 ? 1 : 2);
      if (rx != ry) return rx < ry;
      return x.idx < y.idx;
    });
    // Self-check of the order (the property the dead-lock argument rests on, verified instead of trusted): `placed` = position in
    // the visiting order at which a tile / a PRE slot / a column's inverse is produced; everything a task or a chain column
    // waits for must have been placed before it. A violation (a tie, a NaN in the model after some future change) does not
    // become a hung launch: the structure keeps the launch-per-panel schedule and says so.
    std::vector<int> placed_tile((size_t)nt, -1), placed_pre((size_t)2 * nb, -1), placed_col(nb, -1), placed_begin(nb, -1);
    bool order_ok = true;
    int position = 0;
    auto produced = [&](int where) { if (where < 0) order_ok = false; };
    for (const Event& ev : events) {
      ++position;
      if (ev.chain == 2) {  // column j begins: everything it waits for has been placed; its own panel tile is produced from here on
        const int j = ev.idx, n = seg_of_tile[j];
        const bool first = j == nodes[n].begin;
        if (first) { for (int c : children[n]) produced(placed_col[nodes[c].end - 1]); }
        else produced(placed_col[j - 1]);
        if (chain_info[j] & 1) produced(placed_pre[2 * j]);
        if (chain_info[j] & 2) produced(placed_pre[2 * j + 1]);
        placed_begin[j] = position;
        if (!first) placed_tile[tile_id[(size_t)j * nb + (j - 1)]] = position;
        continue;
      }
      if (ev.chain == 1) {  // column j ends: its inverse is published
        produced(placed_begin[ev.idx]);
        placed_col[ev.idx] = position;
        continue;
      }
      auto tile_there = [&](int r, int k) { produced(placed_tile[tile_id[(size_t)r * nb + k]]); };
      const CholTask& c = gen[ev.idx].t;
      for (int u = c.ub; u < c.ue; ++u) {
        tile_there(c.i, upd[u]);
        if (c.i != c.j) tile_there(c.j, upd[u]);
      }
      if (c.kind == CHOL_TASK_TILE) { produced(placed_col[c.j]); placed_tile[tile_id[(size_t)c.i * nb + c.j]] = position; }
      else placed_pre[c.kind == CHOL_TASK_PRE_DIAG ? 2 * c.j : 2 * c.i + 1] = position;
    }
    O.ok = order_ok;
  };
  // The launch on `helpers` work-groups, the tasks visited in the order O (pre_pool > 0: the first pre_pool helpers take the
  // PRE tasks - what the chains wait for directly - and nothing else).
  auto run_pass = [&](const Order& O, Pass& P, int pre_pool) {
    Times& act = P.act;
    act.L.assign((size_t)nt, 0.0); act.col_begin.assign(nb, 0.0); act.col_fin.assign(nb, 0.0); act.pre.assign((size_t)2 * nb, 0.0);
    std::vector<double> free_at(helpers, 0.0);
    // The helpers of a pool kept sorted by (free time ascending, index DESCENDING) in one small array: the two choices below -
    // the helper that is free latest among those free by `release`, else the one that is free first, the lowest index among
    // equals both times - are the last element of a prefix and the last element of the first group of equals; the chosen
    // helper's new time moves it up by one rotation (instead of a scan of all helpers per task).
    struct FreeKey { double t; int w; };
    auto free_less = [](const FreeKey& a, const FreeKey& b) { return a.t != b.t ? a.t < b.t : a.w > b.w; };
    std::vector<FreeKey> pool_free[2];  // [0]: the PRE pool (helpers [0, pre_pool)), [1]: the others
    for (int w = helpers - 1; w >= 0; --w) pool_free[w < pre_pool ? 0 : 1].push_back(FreeKey{0.0, w});
    double work_us = 0.0, occupied_us = 0.0;
    P.placed.clear();
    P.placed.reserve(gen.size());
    for (const Event& ev : O.events) {
      if (ev.chain == 2) { chain_begin(act, ev.idx); continue; }
      if (ev.chain == 1) { chain_end(act, ev.idx); continue; }
      const CholTask& t = gen[ev.idx].t;
      const int n_upd = t.ue - t.ub;
      const double release = std::max(0.0, O.upd_end[ev.idx] - cU * n_upd - 10.0);
      std::vector<FreeKey>& pool = pool_free[pre_pool > 0 && t.kind != CHOL_TASK_TILE ? 0 : 1];
      // latest free time <= release, lowest index among equals: the element before the first one that is free later
      auto it = std::upper_bound(pool.begin(), pool.end(), FreeKey{release, -1}, free_less);
      // (none is free by then: the earliest, lowest index among equals)
      if (it == pool.begin()) it = std::upper_bound(pool.begin(), pool.end(), FreeKey{pool.front().t, -1}, free_less);
      --it;
      const int best = it->w;
      double f = run_updates_at(act, t.i, t.j, upd.data() + t.ub, (size_t)n_upd, free_at[best]);
      if (t.kind == CHOL_TASK_TILE) {
        f = std::max(f, act.col_fin[t.j]) + cS;
        act.L[tile_id[(size_t)t.i * nb + t.j]] = f;
      } else {
        f += cP;
        act.pre[t.kind == CHOL_TASK_PRE_DIAG ? 2 * t.j : 2 * t.i + 1] = f;
      }
      work_us += cU * n_upd + (t.kind == CHOL_TASK_TILE ? cS : cP);
      occupied_us += f - free_at[best];
      free_at[best] = f;
      {  // its new place: times only grow, so it moves towards the end
        const FreeKey moved{f, best};
        auto to = std::upper_bound(it + 1, pool.end(), moved, free_less);
        std::move(it + 1, to, it);
        *(to - 1) = moved;
      }
      P.placed.push_back({best, ev.idx});
    }
    if (sched_debug)  // (C5: 283 ms of work, 374 ms occupied - 90 ms waiting inside tasks -, 252 helpers x 1.99 ms = 501 ms)
      std::fprintf(stderr, "mavba:   %d helpers for the PRE tasks: work %.0f us, occupied %.0f us, helpers x forward %.0f us\n", pre_pool, work_us,
                   occupied_us, helpers * forward_of(act));
    P.ok = O.ok;
    P.forward = forward_of(act);
  };
  // The PRE tasks are what the chains wait for directly. Besides the shared pool (0) a few sizes of a pool of helpers that take
  // ONLY them are simulated and the shortest launch is kept (C3: no difference from 32 helpers on, shared pool kept; C5: 32 helpers
  // for the 345 PRE tasks, 1 990 -> 1 938 us; a fixed-point iteration of the passes - start times from the previous pass's
  // launch instead of the unlimited one - was tried on the real C5 structure and made it worse: 2 083, 2 059 us).
  Pass best_pass;
  {
    int npre = 0;
    for (const Gen& g : gen) npre += g.t.kind != CHOL_TASK_TILE;
    // (the candidates are independent simulations: one host thread each where a pass is long enough to pay for waking the
    // workers - C5: ~1 ms per pass, C3: 0.07 ms -; the choice among them goes in the fixed order)
    Order order;
    auto run_candidates = [&](const std::vector<int>& pools, const OrderKeys* keys, std::vector<Pass>& cand) {
      cand.assign(pools.size(), Pass{});
      make_order(ideal, keys, order);
      const int T = gen.size() >= 2000 ? std::max(1, std::min(host_threads(), (int)pools.size())) : 1;
      host_run(T, [&](int t) {
        for (size_t q = (size_t)t; q < pools.size(); q += (size_t)T) {
          const int pool = pools[q];
          if (pool > 0 && (pool >= helpers - 1 || pool > npre || npre == (int)gen.size())) continue;
          run_pass(order, cand[q], pool);
        }
      });
    };
    const std::vector<int> pools = {0, 4, 8, 16, 32, 64, 96};
    std::vector<Pass> cand;
    run_candidates(pools, nullptr, cand);
    for (size_t q = 0; q < pools.size(); ++q) {
      Pass& P = cand[q];
      if (!P.ok) continue;
      if (sched_debug) std::fprintf(stderr, "mavba: schedule with %d helpers for the PRE tasks: forward %.1f us\n", pools[q], P.forward);
      if (!best_pass.ok || P.forward < best_pass.forward) best_pass = std::move(P);
    }
    // Where the helpers are the bottleneck (the launch takes much longer than on unlimited helpers) the ORDER of the visit matters:
    // by earliest finish a tile whose consumers are far away is placed as early as one the chain is about to wait for. The
    // second candidate visits by LATEST finish - the time by which a task has to end so that the unlimited launch still ends on
    // time, from a backward pass over the same dependencies (a consumer's update q has to start (n - q) updates + its solve
    // before the consumer's own latest finish). Also a topological order (a producer's latest finish lies before its consumers').
    // C5 (simulated on the real structure, tests/test_chol_schedule.py): 1 938 -> 1 763 us; C3: no difference, first candidate kept.
    if (best_pass.ok && best_pass.forward > 1.3 * forward_of(ideal)) {
      const double E = forward_of(ideal), INF = 1e300;
      std::vector<double> need_tile((size_t)nt, INF), need_pre((size_t)2 * nb, INF), need_col(nb, INF);
      OrderKeys K;
      K.task.assign(gen.size(), E); K.begin.assign(nb, E); K.fin.assign(nb, E);
      struct Item { double t; int rank; int idx; };  // rank as in the visit: column ends (0), tasks (1), column begins (2)
      std::vector<Item> items;
      for (size_t g = 0; g < gen.size(); ++g) {
        const CholTask& t = gen[g].t;
        items.push_back(Item{t.kind == CHOL_TASK_TILE ? ideal.L[tile_id[(size_t)t.i * nb + t.j]] : ideal.pre[t.kind == CHOL_TASK_PRE_DIAG ? 2 * t.j : 2 * t.i + 1], 1, (int)g});
      }
      for (int j = 0; j < nb; ++j) { items.push_back(Item{ideal.col_fin[j], 0, j}); items.push_back(Item{ideal.col_begin[j], 2, j}); }
      std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) {  // reverse of the visiting order
        if (x.t != y.t) return x.t > y.t;
        if (x.rank != y.rank) return x.rank > y.rank;
        return x.idx > y.idx;
      });
      for (const Item& it : items) {
        if (it.rank == 1) {
          const CholTask& t = gen[it.idx].t;
          const double lf = std::min(E, t.kind == CHOL_TASK_TILE ? need_tile[tile_id[(size_t)t.i * nb + t.j]]
                                                                  : need_pre[t.kind == CHOL_TASK_PRE_DIAG ? 2 * t.j : 2 * t.i + 1]);
          K.task[it.idx] = lf;
          const double tail = t.kind == CHOL_TASK_TILE ? cS : cP;
          if (t.kind == CHOL_TASK_TILE) need_col[t.j] = std::min(need_col[t.j], lf - cS);
          const int n = t.ue - t.ub;
          for (int q = 0; q < n; ++q) {
            const double ls = lf - tail - cU * (n - q);
            const int k = upd[t.ub + q];
            double& a = need_tile[tile_id[(size_t)t.i * nb + k]];
            a = std::min(a, ls);
            if (t.i != t.j) { double& b = need_tile[tile_id[(size_t)t.j * nb + k]]; b = std::min(b, ls); }
          }
        } else if (it.rank == 0) {
          const int j = it.idx, n = seg_of_tile[j];
          double lf = std::min(E, need_col[j]);
          if (j + 1 < nodes[n].end) lf = std::min(lf, K.begin[j + 1]);
          else if (nodes[n].parent >= 0) lf = std::min(lf, K.begin[nodes[nodes[n].parent].begin]);
          K.fin[j] = lf;
        } else {
          const int j = it.idx;
          const bool first = j == nodes[seg_of_tile[j]].begin;
          double lb = K.fin[j] - (first ? cColFirst : cCol);
          if (!first) lb = std::min(lb, need_tile[tile_id[(size_t)j * nb + (j - 1)]] - cSub);
          K.begin[j] = lb;
          need_pre[2 * j] = std::min(need_pre[2 * j], lb);
          need_pre[2 * j + 1] = std::min(need_pre[2 * j + 1], lb);
        }
      }
      const std::vector<int> pools2 = {0, 32};
      run_candidates(pools2, &K, cand);
      for (size_t q = 0; q < pools2.size(); ++q) {
        Pass& P = cand[q];
        if (!P.ok) continue;
        if (sched_debug) std::fprintf(stderr, "mavba: latest-finish order, %d helpers for the PRE tasks: forward %.1f us\n", pools2[q], P.forward);
        if (P.forward < 0.98 * best_pass.forward) best_pass = std::move(P);
      }
    }
  }
  lap("list scheduling passes");
  const bool order_ok = best_pass.ok;
  std::vector<std::vector<CholTask>> helper_tasks(helpers);  // the helpers' queues of the pass that is kept, in visiting order
  for (const auto& hp : best_pass.placed) helper_tasks[hp.first].push_back(gen[hp.second].t);
  Times& act = best_pass.act;
  if (!order_ok) {
    std::fprintf(stderr, "mavba: the persistent factorisation's task order failed its self-check; using the launch-per-panel schedule\n");
    return hipSuccess;
  }
  int used = 0;
  for (int w = 0; w < helpers; ++w) used += !helper_tasks[w].empty();
  const int grid = nch + used;
  predicted_forward_us = 0.0;
  for (int j = 0; j < nb; ++j) predicted_forward_us = std::max(predicted_forward_us, act.col_fin[j]);
  for (size_t id = 0; id < act.L.size(); ++id) predicted_forward_us = std::max(predicted_forward_us, act.L[id]);  // (the last panel solves)
  // Persistent or launch-per-panel? Rounds 2-3 drew the line at 100 tile updates per helper (C5's 75 582 updates then took
  // 4.96 ms persistent against 3.45 ms). With the model the question is asked directly: the launch-per-panel schedule costs a
  // step about max(18 us, 8 us + 0.026 us per 64^3 tile update) (C3: 24 steps, ~0.55 ms; C5: 69 steps, 82 k updates, 2.6 ms
  // measured); the persistent launch what the simulation above says (C5: 1.99 ms predicted, 2.08 measured - now the faster one).
  double lpp_us = 0.0;
  for (const CholStep& S : steps) lpp_us += S.kind != 0 ? 12.0 : std::max(18.0, 8.0 + 0.026 * (double)S.tasks);
  lpp_estimate_us = lpp_us;
  if (mode == 1 && predicted_forward_us > lpp_us && !host_only) return hipSuccess;
  std::vector<std::vector<CholTask>> wg_tasks(grid);
  for (int n = 0; n < nseg; ++n) wg_tasks[chain_wg[n]].push_back(CholTask{CHOL_TASK_CHAIN, nodes[n].begin, nodes[n].end, 0, 0});
  {
    int base = nch;
    for (int w = 0; w < helpers; ++w)
      if (!helper_tasks[w].empty()) wg_tasks[base++] = helper_tasks[w];
  }
  std::vector<CholTask> tasks;
  std::vector<int> wg_begin(grid + 1, 0);
  for (int w = 0; w < grid; ++w) {
    wg_begin[w] = (int)tasks.size();
    tasks.insert(tasks.end(), wg_tasks[w].begin(), wg_tasks[w].end());
  }
  wg_begin[grid] = (int)tasks.size();
  // ---- device copies ----
  std::vector<int> pack(wg_begin);
  const size_t o_upd = pack.size();
  pack.insert(pack.end(), upd.begin(), upd.end());
  const size_t o_tid = pack.size();
  pack.insert(pack.end(), tile_id.begin(), tile_id.end());
  const size_t o_ci = pack.size();
  pack.insert(pack.end(), chain_info.begin(), chain_info.end());
  const size_t nflags = (size_t)nt + 3 * (size_t)nb + 1;
  lap("queues");
  if (host_only) {  // (the test entry reads the schedule from these)
    h_tasks = tasks; h_wg_begin = wg_begin; h_upd = upd; h_chain_info = chain_info;
    persist_grid = grid; persist_chain_wgs = nch; persist_tiles = nt; persist_updates = nupd;
    persist_ok = true;
    return hipSuccess;
  }
  hipError_t e = device_alloc(reinterpret_cast<void**>(&d_pints), pack.size() * sizeof(int));
  if (e == hipSuccess) e = copy_h2d_staged(d_pints, pack.data(), pack.size() * sizeof(int), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_tasks), tasks.size() * sizeof(CholTask));
  if (e == hipSuccess) e = copy_h2d_staged(d_tasks, tasks.data(), tasks.size() * sizeof(CholTask), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_pflags), nflags * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemsetAsync(d_pflags, 0, nflags * sizeof(unsigned), st);
  if (e == hipSuccess) e = device_alloc(reinterpret_cast<void**>(&d_pre), (size_t)2 * nb * NB * NB * sizeof(double));
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  release_staged(st);
  if (e != hipSuccess) return e;
  if (std::getenv("MAVBA_CHOL_TRACE")) {
    std::fprintf(stderr, "mavba: persistent factorisation, timing model: %.1f us forward on %d helpers (%.1f us on unlimited helpers)\n",
                 predicted_forward_us, used, *std::max_element(ideal.col_fin.begin(), ideal.col_fin.end()));
    h_tasks = tasks; h_wg_begin = wg_begin;
    const size_t ntr = (size_t)8 * nb + (size_t)4 * tasks.size();
    if (device_alloc(reinterpret_cast<void**>(&d_trace), ntr * 8) == hipSuccess) (void)hipMemset(d_trace, 0, ntr * 8);
    else d_trace = nullptr;
  }
  d_wg_begin = d_pints; d_upd = d_pints + o_upd; d_tile_id = d_pints + o_tid; d_chain_info = d_pints + o_ci;
  persist_grid = grid; persist_chain_wgs = nch; persist_tiles = nt; persist_updates = nupd;
  epoch = 0;
  persist_ok = true;
  return hipSuccess;
}

// Systems of one or two tile columns (n <= 128: a local bundle-adjustment window of up to ~20 images): ONE work-group
// factorises, substitutes forward and backward with everything in LDS - no hand-offs between work-groups, one launch
// instead of two (the persistent launch + the backward substitution cost 43 us for the 128 x 128 system of a 10-image
// window; this is the same arithmetic: L00, L10 = A10 L00^-T, A11 - L10 L10^T, L11, then the two triangular solves with
// the inverted diagonal tiles). M is only read.
// UPD (round 4): the work-group goes on to the camera update of the LM step (k_update_cameras' work: it reads the solution
// this work-group has just written) - one launch less on the 75 us iteration of a local window.
struct SmallSlots { int s00, s10, s11, r0, r1; };  // tiles (0,0), (1,0), (1,1) and the right-hand side's two tiles; < 0: not in the store (zero)
__device__ __forceinline__ void load_tile_or_zero(const double* __restrict__ M, int slot, double* S, int tid) {
  if (slot >= 0) { load_tile(M + ((size_t)slot << 12), NB, S, tid); return; }
  for (int e = tid; e < NB * NB; e += 256) S[(e >> 6) * GLD + (e & 63)] = 0.0;
}
template <bool UPD>
__global__ void __launch_bounds__(256) k_chol_small(const double* __restrict__ M, SmallSlots sl, int nb_all, int nb,
                                                    double* __restrict__ fail, double* __restrict__ y,
                                                    const int* __restrict__ scatter, double* y_nat, CamUpdateArgs U) {
  // nb (1 or 2): the leading tile columns that hold free parameters; columns beyond them (up to nb_all tiles) are unit
  // diagonal with a zero right-hand side: their solution is 0
  __shared__ __attribute__((aligned(16))) double W[NB * GLD];
  __shared__ __attribute__((aligned(16))) double I0[NB * GLD];
  __shared__ __attribute__((aligned(16))) double I1[NB * GLD];
  __shared__ __attribute__((aligned(16))) double P[NB * GLD];
  __shared__ double vv[2 * NB], zz[2 * NB], xx[2 * NB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
  load_tile_or_zero(M, sl.s00, W, tid);
  if (nb > 1) load_tile_or_zero(M, sl.s10, P, tid);
  if (tid < nb * NB) { const int rs = tid < NB ? sl.r0 : sl.r1; vv[tid] = rs >= 0 ? M[((size_t)rs << 12) + (tid & 63)] : 0.0; }
  __syncthreads();
  bool ok = tile_factor_inverse(W, I0, tid);
  __syncthreads();
  auto lower_mv = [&](const double* Li, const double* in, double* out) {  // out = Li in   (Li lower, upper part zero)
    if (wv == 0) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int j = 0; j < NB; j += 2) { s0 = __builtin_fma(Li[lane * GLD + j], in[j], s0); s1 = __builtin_fma(Li[lane * GLD + j + 1], in[j + 1], s1); }
      out[lane] = s0 + s1;
    }
  };
  auto lower_tmv = [&](const double* Li, const double* in, double* out) {  // out = Li^T in
    if (wv == 0) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int i = 0; i < NB; i += 2) { s0 = __builtin_fma(Li[i * GLD + lane], in[i], s0); s1 = __builtin_fma(Li[(i + 1) * GLD + lane], in[i + 1], s1); }
      out[lane] = s0 + s1;
    }
  };
  lower_mv(I0, vv, zz);  // z0 = L00^-1 v0
  if (nb > 1) {
    d4 acc[2][2];
    mfma_quadrant_nt(P, I0, wr, wc, lane, acc);  // L10 = A10 (L00^-1)^T
    load_tile_or_zero(M, sl.s11, W, tid);  // A11 (W's L00 is not needed any more)
    __syncthreads();
    quadrant_to_lds(P, wr, wc, lane, acc);
    __syncthreads();
    d4 upd[2][2], cur[2][2];
    mfma_quadrant_nt(P, P, wr, wc, lane, upd);  // L10 L10^T
    quadrant_from_lds(W, wr, wc, lane, cur);
    quadrant_sub(cur, upd);
    if (wv == 0) {  // v1 - L10 z0
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int j = 0; j < NB; j += 2) { s0 = __builtin_fma(P[lane * GLD + j], zz[j], s0); s1 = __builtin_fma(P[lane * GLD + j + 1], zz[j + 1], s1); }
      xx[NB + lane] = vv[NB + lane] - (s0 + s1);
    }
    __syncthreads();
    quadrant_to_lds(W, wr, wc, lane, cur);
    __syncthreads();
    ok = tile_factor_inverse(W, I1, tid) && ok;
    __syncthreads();
    lower_mv(I1, xx + NB, zz + NB);   // z1
    __syncthreads();
    lower_tmv(I1, zz + NB, xx + NB);  // x1 = L11^-T z1
    __syncthreads();
    if (wv == 0) {  // z0 - L10^T x1
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
      for (int i = 0; i < NB; i += 2) { s0 = __builtin_fma(P[i * GLD + lane], xx[NB + i], s0); s1 = __builtin_fma(P[(i + 1) * GLD + lane], xx[NB + i + 1], s1); }
      zz[lane] -= s0 + s1;
    }
  }
  __syncthreads();
  lower_tmv(I0, zz, xx);  // x0
  __syncthreads();
  if (tid == 0 && !ok) atomicAdd(fail, 1.0);
  for (int c = tid; c < nb_all * NB; c += 256) {
    const double x = c < nb * NB ? xx[c] : 0.0;
    y[c] = x;
    if (scatter) { const int t = scatter[c]; if (t >= 0) y_nat[t] = x; }
  }
  if (UPD) {
    __shared__ double s_red[4];
    __syncthreads();  // (the solution in the variables' order is in memory, for every lane of this work-group)
    const int groups = (U.NI + kUpdImagesPerGroup - 1) / kUpdImagesPerGroup;
    for (int vb = 0; vb < (groups > 0 ? groups : 1); ++vb) {
      update_cameras_body(vb, U.NI, U.NC, U.cam_part, U.radius, U.dmin, U.dmax, y_nat, U.scale_cam, U.img_rec, U.cam_rec, U.poses, U.intr,
                          U.cand_poses, U.cand_intr, U.delta_cam, U.partial3, U.cand_camrec, s_red);
      __syncthreads();
    }
  }
}

// diag_ws: n_pad * 64 doubles (the inverses of the factor's diagonal tiles); L: second
// (n_pad + 64) x n_pad matrix receiving the factor's off-diagonal tiles and the
// forward-substituted right-hand side. `cs` = tile structure + launch schedule of the matrix.
static bool chol_small_path() {
  static const bool on = [] { const char* e = std::getenv("MAVBA_CHOL_SMALL"); return !e || std::atoi(e) != 0; }();
  return on;
}
bool dense_spd_solve_is_small(int n_pad, const CholStructure& cs) {
  const int nb = n_pad / NB;
  const int nb_active = cs.active_tiles > 0 ? std::min(cs.active_tiles, nb) : nb;
  return nb_active <= 2 && chol_small_path();
}
bool dense_spd_solve_device(hipStream_t st, double* M, int n_pad, double* y, double* fail,
                            double* diag_ws, double* L, const CholStructure& cs,
                            const int* y_scatter, double* y_nat, bool allow_persistent, hipEvent_t after_factor,
                            const CamUpdateArgs* upd) {
  const int nb = n_pad / NB;
  const int* slot = cs.d_tile_slot;
  double* inv = diag_ws;
  const int nb_active = cs.active_tiles > 0 ? std::min(cs.active_tiles, nb) : nb;
  if (dense_spd_solve_is_small(n_pad, cs)) {
    const bool with_update = upd != nullptr && y_scatter != nullptr && y_nat != nullptr;
    auto hs = [&](int i, int k) { return i <= nb && k < nb ? cs.tile_slot[(size_t)i * nb + k] : -1; };
    const SmallSlots sl{hs(0, 0), nb > 1 ? hs(1, 0) : -1, nb > 1 ? hs(1, 1) : -1, hs(nb, 0), nb > 1 ? hs(nb, 1) : -1};
    if (with_update)
      hipLaunchKernelGGL(k_chol_small<true>, dim3(1), dim3(256), 0, st, M, sl, nb, nb_active, fail, y, y_scatter, y_nat, *upd);
    else
      hipLaunchKernelGGL(k_chol_small<false>, dim3(1), dim3(256), 0, st, M, sl, nb, nb_active, fail, y, y_scatter, y_nat, CamUpdateArgs{});
    if (after_factor) (void)hipEventRecord(after_factor, st);  // (one launch does both halves)
    return with_update;
  }
  const unsigned epoch = ++cs.epoch;  // flags of this solve (forward hand-offs and backward substitution)
  if (allow_persistent && cs.persist_ok) {
    CholPersistArgs A;
    A.M = M; A.L = L; A.inv = inv; A.pre = cs.d_pre; A.nb = nb;
    A.tasks = cs.d_tasks; A.wg_begin = cs.d_wg_begin; A.upd = cs.d_upd; A.tile_id = cs.d_tile_id; A.chain_info = cs.d_chain_info;
    A.lflag = cs.d_pflags; A.dflag = cs.d_pflags + cs.persist_tiles; A.pflag = A.dflag + nb; A.abort_flag = A.pflag + 2 * nb;
    A.epoch = epoch; A.fail = fail; A.trace = cs.d_trace;
    // (Round 5 built the backward substitution as the tail of this launch - no second launch, no gap -, verified it bit-identical
    // and measured it SLOWER: inside the launch every factor tile has to be read with system-scope loads, C3 0.265 + 0.036 ->
    // 0.310 + 0.005 ms, and the tail's mere presence cost the forward pass 1-2 us. The code is in scripts/_dbg/pruned_r06.patch.)
    static const int drop = [] { const char* e = std::getenv("MAVBA_CHOL_TEST_DROP_WG"); return e ? std::atoi(e) : -1; }();
    A.drop_wg = drop;
    // Two persistent launches must never share the device (each needs every CU for its resident grid). Launches on ONE stream
    // are ordered by the stream; as soon as a second stream of this process launches one (sessions on other threads), the
    // device is drained once and from then on every launch waits for the previous one's event and records its own. A process
    // that only ever uses one stream per device - MAVMAP's serial calls: the stream cache hands the same one to every session -
    // never pays for the event: recorded between this launch and the back-substitution it cost 4-5 us per solve (round 5,
    // A/B in one visit: C2 3 891 -> 3 954 iter/s). The remembered stream is only compared, never used.
    {
      static std::mutex chain_m;
      static hipEvent_t last[64] = {};
      static hipStream_t only_stream[64] = {};
      static bool chained[64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      const bool tracked = dev >= 0 && dev < 64;
      std::lock_guard<std::mutex> g(chain_m);
      if (tracked && !chained[dev]) {
        if (!only_stream[dev]) only_stream[dev] = st;
        else if (only_stream[dev] != st) { chained[dev] = true; (void)hipDeviceSynchronize(); }
      }
      const bool chain = tracked && chained[dev];
      if (chain && last[dev]) (void)hipStreamWaitEvent(st, last[dev], 0);
      hipLaunchKernelGGL(k_chol_persist, dim3(cs.persist_grid), dim3(256), 0, st, A);
      if (chain) {
        if (!last[dev]) (void)hipEventCreateWithFlags(&last[dev], hipEventDisableTiming);
        if (last[dev]) (void)hipEventRecord(last[dev], st);
      }
    }
  } else {
  if (cs.shadow_doubles) {
    // the shadow blocks of the concurrent fronts - dense over the ancestors' tile range, far larger than the tile store itself -
    // exist only once this schedule really runs (round 6; rounds 3-5 allocated them with the structure: 5 GB at C5 that the
    // persistent launch never touched)
    if (!cs.d_shadow && device_alloc(reinterpret_cast<void**>(&cs.d_shadow), cs.shadow_doubles * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      cs.d_shadow = nullptr;
      static const double kNoMemory = 1e30;  // (reported like a time-out of the persistent launch: the step is invalid, nothing is read from a null block)
      std::fprintf(stderr, "mavba: out of device memory for the launch-per-panel factorisation's shadow blocks (%.1f GB)\n", cs.shadow_doubles * 8e-9);
      (void)hipMemcpyAsync(fail, &kNoMemory, sizeof(double), hipMemcpyHostToDevice, st);
      if (after_factor) (void)hipEventRecord(after_factor, st);
      return false;
    }
    (void)hipMemsetAsync(cs.d_shadow, 0, cs.shadow_doubles * sizeof(double), st);
  }
  static const int fuse_below = [] { const char* e = std::getenv("MAVBA_CHOL_FUSE"); return e ? std::atoi(e) : kFuseBelow; }();  // tuning knobs
  static const int fuse_tasks = [] { const char* e = std::getenv("MAVBA_CHOL_FUSE_TASKS"); return e ? std::atoi(e) : kFuseTasks; }();
  hipLaunchKernelGGL(k_chol_diag0, dim3(cs.num_leaf_init), dim3(256), 0, st, M, slot, nb, cs.d_init, inv, fail, cs.d_flags, nb);
  for (const CholStep& S : cs.steps) {
    if (S.kind == 1) {
      const int nt = nb - S.merge_begin;
      if (S.nf > 0)
        hipLaunchKernelGGL(k_chol_merge, dim3(nt, nt + 1), dim3(256), 0, st, M, slot, nb, cs.d_shadow, cs.d_merges + S.front_off, S.nf,
                           S.merge_begin, nb);
      hipLaunchKernelGGL(k_chol_diag0, dim3(S.nf0), dim3(256), 0, st, M, slot, nb, cs.d_init + S.init_off, inv, fail, cs.d_flags, 0);
      continue;
    }
    const CholFront* F = cs.d_fronts + S.front_off;
    if (S.max_na > fuse_below || S.tasks > fuse_tasks) {
      // much trailing work (more tile updates than one round of the CUs absorbs): one panel solve, then a lean
      // update (one product per work-group, 2 work-groups per CU)
      hipLaunchKernelGGL(k_chol_trsm, dim3(S.max_na + 1, 1, S.nf), dim3(256), 0, st, M, L, slot, nb, F, inv, cs.d_rows, nb);
      hipLaunchKernelGGL((k_chol_update<false>), dim3(S.max_na, S.max_na + 1, S.nf), dim3(256), 0, st, M, L, slot, nb, F, inv, fail,
                         cs.d_rows, nb, cs.d_shadow);
    } else if (S.max_na > 0) {
      // small trailing matrix: latency matters, fold the panel solve into the update launch
      hipLaunchKernelGGL((k_chol_update<true>), dim3(S.max_na, S.max_na + 1, S.nf), dim3(256), 0, st, M, L, slot, nb, F, inv, fail,
                         cs.d_rows, nb, cs.d_shadow);
    }
    if (S.nf0)  // fronts with nothing below their tile: only the right-hand-side block is left
      hipLaunchKernelGGL(k_chol_trsm, dim3(1, 1, S.nf0), dim3(256), 0, st, M, L, slot, nb, F + S.nf, inv, cs.d_rows, nb);
  }
  }
  if (after_factor) (void)hipEventRecord(after_factor, st);
  if (nb <= kMaxBacksolveGroups) {
    const int cus = device_cu_count();
    hipLaunchKernelGGL(k_chol_backsolve_all, dim3(std::min(nb, cus > 0 ? 2 * cus : 64)), dim3(256), 0, st, L, slot, nb, cs.nseg, cs.d_seg_of_tile, cs.d_seg_first,
                       inv, y, cs.d_flags, epoch, y_scatter, y_nat);
  } else {
    // more tile rows than work-groups that are certainly resident: one small launch per tile (single segment)
    for (int k = nb - 1; k >= 0; --k)
      hipLaunchKernelGGL(k_chol_backsolve, dim3(k - cs.seg_first[k] + 1), dim3(64), 0, st, L, slot, nb, k, cs.seg_first[k], inv, L, y);
    if (y_scatter) hipLaunchKernelGGL(k_scatter_y, dim3((n_pad + 255) / 256), dim3(256), 0, st, n_pad, y_scatter, y, y_nat);
  }
  return false;
}

}  // namespace mavba
