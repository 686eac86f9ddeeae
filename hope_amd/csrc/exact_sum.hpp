// exact_sum.hpp -- the reference's SEQUENTIAL float sums, evaluated in parallel, bit for bit.
//
// getCloudZInfo (utilities.cpp:751-769), calNormalMean (utilities.cpp:257-282) and the sums of
// normalAnalysis (utilities.cpp:151-255) add the members of a cluster one after the other in fp32:
//     s = fl(fl(fl(a0 + a1) + a2) + ...)
// A chain of dependent adds costs one wave ~3.4 ns per member (kernels.hpp, chain_sum): 2 ms per pass
// over a 570 000-member floor, and no partition of the cluster shortens it.  This file removes the
// chain without changing one bit of the result.
//
// While the running sum s stays inside one binade [2^e, 2^(e+1)) its spacing u = 2^(e-23) is fixed,
// s = S*u with an integer S in [2^23, 2^24), and round-to-nearest-even of s + a is
//     S' = RNE(S + a/u) = S + q,   q = floor(a/u) (+1 if the fraction of a/u is above a half;
//                                                  on an exact half: whichever makes S' even)
// -- INTEGER arithmetic, hence associative.  Only an exact half looks at the state, and only at its
// parity; after it S' is even whatever came before.  So for a block of 64 members (one per lane) and a
// PREDICTED exponent e, every q is computed independently (parities from two ballots), and the block
// collapses into a record: {e, Q[parity of S at entry], the range of S at entry for which every
// intermediate sum stays strictly inside the binade}.  Sixteen records compose into one for 1024 members.
// The chain then walks records with scalar instructions: a compare and an add per 1024 members where
// it was 1024 dependent v_add_f32.  The exponent is predicted from a double-precision prefix sum
// (off by the rounding the chain accumulates, so wrong only near a binade boundary).  Wherever a record
// does not apply (the binade changes, the prediction is off, s is still small against its terms, a term
// is infinite or NaN) the wave evaluates that block exactly: integer steps as far as they hold, one real
// v_add_f32 where they stop, and plain sequential adds when integer steps accept nothing twice in a row
// (terms that cancel: the worst case is the old chain).
//
// Exactness argument, for the record (tests/test_exact_sum.py holds it against numpy's sequential
// float32 sum on adversarial inputs): with |S| in [2^23, 2^24) and A = a/u, fl(s + a) = u*RNE(S + A)
// whenever 2^23 <= |S + A| <= 2^24, because that is the binade whose spacing is u.  Every accepted step
// has 2^23 + 1 <= |S'| <= 2^24 - 1, which implies it.  A term enters only if |a| < 2^(e-1) (at least two
// fraction bits below u), so |q| <= 2^22 and no partial sum overflows an int.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace xs {

constexpr int kBlk = 64;             // members per block: one per lane
constexpr int kSub = 16;             // blocks per super block
constexpr int kSuper = kBlk * kSub;  // 1024 members
constexpr int kNoInt = INT_MIN;      // chain state: s has no integer form (zero, denormal, infinite, NaN)
constexpr int kInvalid = INT_MAX;    // record: the block has no integer form under its predicted exponent
constexpr int kLoS = (1 << 23) + 1;  // every intermediate |S| stays in [kLoS, kHiS]
constexpr int kHiS = (1 << 24) - 1;
constexpr int kPMax = 1 << 25;       // a block whose partial sums leave (-kPMax, kPMax) can never apply

struct RecA {
  int e, q0, q1, lo;
};  // + hi in an array of its own: S in [lo, hi] at entry, Q = (S & 1) ? q1 : q0

struct Stats {
  unsigned slow_blocks;   // blocks evaluated by the wave (a record did not apply)
  unsigned float_steps;   // single v_add_f32 steps inside them
  unsigned plain_blocks;  // blocks finished by plain sequential adds
  unsigned top_hits;      // super-block records applied
};

__device__ __forceinline__ float readlane_f(float v, int lane)  // (the builtin is typed int: convert the BITS)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ unsigned long long lanemask_lt()
{
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// a/u for u = 2^eu: integer part I = floor(a/u), and where the fraction stands against a half.
// valid: the term may enter an integer step (zero, or at least two fraction bits below u).
__device__ __forceinline__ void decompose(float a, int eu, bool& valid, int& I, bool& up, bool& tie)
{
  const unsigned bits = __float_as_uint(a);
  const int bexp = (int)((bits >> 23) & 0xffu);
  const int frac = (int)(bits & 0x7fffffu);
  const int m = bexp ? (frac | 0x800000) : frac;
  const int xa = (bexp ? bexp : 1) - 150;  // a = +-m * 2^xa
  const int afx = (bits >> 31) ? -m : m;
  int d = eu - xa;                         // fraction bits of a/u
  valid = (m == 0) || (d >= 2 && bexp != 255);
  d = d < 2 ? 2 : (d > 31 ? 31 : d);
  I = afx >> d;                            // arithmetic shift: floor
  const unsigned fr = (unsigned)afx & ((1u << d) - 1u);
  const unsigned half = 1u << (d - 1);
  up = fr > half;
  tie = fr == half;
}

// q of this lane, given the parity of S in front of the first active lane.  Lanes that are not `use`
// contribute 0 (everything behind the first of them is discarded by the callers).
// first_tie_delta (uniform): Q(parity 1) - Q(parity 0) of the whole block, i.e. +-1 if there is a tie.
__device__ __forceinline__ int lane_q(bool use, int I, bool up, bool tie, int p_in, int* first_tie_delta)
{
  const int base = I + (up ? 1 : 0);
  const unsigned long long T = __builtin_amdgcn_ballot_w64(use && tie);
  const unsigned long long O = __builtin_amdgcn_ballot_w64(use && !tie && (base & 1));
  const unsigned long long lt = lanemask_lt();
  const unsigned long long Tl = T & lt;
  int par;
  if (Tl == 0ull) {
    par = (p_in + __popcll(O & lt)) & 1;
  } else {
    const int t = 63 - __clzll((long long)Tl);  // the last tie before this lane left S even
    par = __popcll(O & lt & ~((2ull << t) - 1ull)) & 1;
  }
  const int x = (par + I) & 1;
  const int q = tie ? I + x : base;
  if (first_tie_delta) {
    int delta = 0;
    if (T != 0ull) {
      const int t0 = __ffsll((long long)T) - 1;
      const int x0 = __builtin_amdgcn_readlane(x, t0);  // computed with p_in: the other parity flips it
      delta = 1 - 2 * x0;
    }
    *first_tie_delta = delta;
  }
  return use ? q : 0;
}

// inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (seven adds; a shuffle through the
// LDS crossbar per step costs the chain's slow path ten times as much)
__device__ __forceinline__ int wave_incl_scan(int v)
{
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);  // row_shr:3   -> sums of 4 within a row of 16
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, true);  // row_shr:4   -> sums of 8
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, true);  // row_shr:8   -> inclusive within the row
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// exponent of a normal, finite, non-zero float; kNoInt otherwise
__device__ __forceinline__ int exponent_of(float s)
{
  const int bexp = (int)((__float_as_uint(s) >> 23) & 0xffu);
  return (bexp == 0 || bexp == 255) ? kNoInt : bexp - 127;
}
__device__ __forceinline__ int mantissa_of(float s)  // signed S with s = S * 2^(e-23)
{
  const unsigned bits = __float_as_uint(s);
  const int m = (int)((bits & 0x7fffffu) | 0x800000u);
  return (bits >> 31) ? -m : m;
}
__device__ __forceinline__ float to_float(int S, int e)  // |S| in [2^23, 2^24)
{
  const unsigned mag = (unsigned)(S < 0 ? -S : S);
  return __uint_as_float((S < 0 ? 0x80000000u : 0u) | ((unsigned)(e + 127) << 23) | (mag & 0x7fffffu));
}

// ---- block sums in double (the predictor's input) --------------------------------------------
// grid (n_super, n_chains), 256 threads: wave w sums blocks 4w..4w+3 of the super block
__global__ void __launch_bounds__(256) k_xs_blocksums(const float* terms, size_t chain_stride, double* bsum,
                                                      size_t bsum_stride)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t sb = blockIdx.x;
  const float* t = terms + blockIdx.y * chain_stride + sb * kSuper;
  double* o = bsum + blockIdx.y * bsum_stride + sb * kSub;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = w * 4 + k;
    double v = (double)t[b * kBlk + lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) o[b] = v;
  }
}

// ---- exclusive prefix of the block sums per (cluster, chain): the predicted s in front of a block ----
struct Span {
  unsigned sb_off;  // first super block
  unsigned n_sb;    // super blocks
};
// grid (n_spans, n_chains), 256 threads; a thread owns the sixteen block sums of one super block
__global__ void __launch_bounds__(256) k_xs_scan(const Span* spans, const double* bsum, size_t bsum_stride, float* spred)
{
  __shared__ double s_w[4];
  __shared__ double s_carry;
  const Span sp = spans[blockIdx.x];
  const size_t base = blockIdx.y * bsum_stride + (size_t)sp.sb_off * kSub;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0.0;
  __syncthreads();
  for (unsigned sb0 = 0; sb0 < sp.n_sb; sb0 += 256) {
    const unsigned sb = sb0 + threadIdx.x;
    double v[kSub];
    double tot = 0.0;
    if (sb < sp.n_sb) {
#pragma unroll
      for (int k = 0; k < kSub; ++k) {
        v[k] = bsum[base + (size_t)sb * kSub + k];
        tot += v[k];
      }
    }
    double inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    double pre = s_carry;
    for (int i = 0; i < w; ++i) pre += s_w[i];
    if (sb < sp.n_sb) {
      double run = pre + (inc - tot);
#pragma unroll
      for (int k = 0; k < kSub; ++k) {
        spred[base + (size_t)sb * kSub + k] = (float)run;
        run += v[k];
      }
    }
    __syncthreads();
    if (threadIdx.x == 255) s_carry = pre + inc;
    __syncthreads();
  }
}

// ---- records ---------------------------------------------------------------------------------
__device__ __forceinline__ void compose(RecA& a, int& a_hi, const RecA& b, int b_hi)
{
  if (a.e == kInvalid || b.e != a.e) {
    a.e = kInvalid;
    return;
  }
  const int qmin = min(a.q0, a.q1), qmax = max(a.q0, a.q1);
  const int q0 = a.q0 + ((a.q0 & 1) ? b.q1 : b.q0);        // entry parity 0: parity after a is that of q0
  const int q1 = a.q1 + (((1 + a.q1) & 1) ? b.q1 : b.q0);  // entry parity 1
  a.lo = max(a.lo, b.lo - qmin);
  a_hi = min(a_hi, b_hi - qmax);
  a.q0 = q0;
  a.q1 = q1;
}

// grid (n_super, n_chains), 256 threads: wave w makes the records of blocks 4w..4w+3, then wave 0
// composes the sixteen into the super block's
__global__ void __launch_bounds__(256) k_xs_records(const float* terms, size_t chain_stride, const float* spred,
                                                    size_t blk_stride, RecA* recA, int* recB, RecA* topA, int* topB,
                                                    size_t top_stride)
{
  __shared__ RecA s_a[kSub];
  __shared__ int s_hi[kSub];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t sb = blockIdx.x;
  const float* t = terms + blockIdx.y * chain_stride + sb * kSuper;
  const size_t blk0 = blockIdx.y * blk_stride + sb * kSub;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int b = w * 4 + k;
    const float sp = spred[blk0 + b];
    const int e = exponent_of(sp);
    RecA r;
    int hi = 0;
    r.e = kInvalid;
    r.q0 = r.q1 = r.lo = 0;
    if (e != kNoInt) {
      bool valid, up, tie;
      int I;
      decompose(t[b * kBlk + lane], e - 23, valid, I, up, tie);
      int delta;
      const int q = lane_q(valid, I, up, tie, 0, &delta);
      const int P = wave_incl_scan(q);
      const int mn = wave_min(P), mx = wave_max(P);
      const int Q0 = __builtin_amdgcn_readlane(P, 63);
      const bool all_valid = __all(valid);
      if (all_valid && mn > -kPMax && mx < kPMax) {
        const int widen = delta != 0 ? 1 : 0;
        r.e = e;
        r.q0 = Q0;
        r.q1 = Q0 + delta;
        if (sp > 0.f) {
          r.lo = kLoS - mn + widen;
          hi = kHiS - mx - widen;
        } else {
          r.lo = -kHiS - mn + widen;
          hi = -kLoS - mx - widen;
        }
      }
    }
    if (lane == 0) {
      s_a[b] = r;
      s_hi[b] = hi;
      recA[blk0 + b] = r;
      recB[blk0 + b] = hi;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RecA a = s_a[0];
    int hi = s_hi[0];
    for (int b = 1; b < kSub; ++b) compose(a, hi, s_a[b], s_hi[b]);
    topA[blockIdx.y * top_stride + sb] = a;
    topB[blockIdx.y * top_stride + sb] = hi;
  }
}

// ---- the chain -------------------------------------------------------------------------------
struct State {  // wave-uniform
  int e;      // exponent, or kNoInt: then s holds the sum
  int S;
  float s;
};
__device__ __forceinline__ void state_from_float(State& st, float s)
{
  st.e = exponent_of(s);
  st.s = s;
  st.S = st.e != kNoInt ? mantissa_of(s) : 0;
}
__device__ __forceinline__ float state_to_float(const State& st)
{
  return st.e != kNoInt ? to_float(st.S, st.e) : st.s;
}
__device__ __forceinline__ bool apply(State& st, int e, int q0, int q1, int lo, int hi)
{
  if (e != st.e || st.S < lo || st.S > hi) return false;  // (kInvalid and kNoInt never meet)
  st.S += (st.S & 1) ? q1 : q0;
  return true;
}

// one block, exactly, by the whole wave (a = this lane's term)
__device__ __forceinline__ void slow_block(State& st, float a, Stats* stats)
{
  const int lane = threadIdx.x & 63;
  // a block of zeros leaves any sum as it is (s is never -0: it starts at +0 and x + (-x) = +0)
  if (__builtin_amdgcn_ballot_w64(a != 0.f) == 0ull) return;
  unsigned n_float = 0;
  int barren = 0;  // steps in a row in which no integer step was accepted
  int start = 0;
  while (start < kBlk) {
    if (barren >= 2) {
      // terms that cancel or dwarf the sum (or a sum that is infinite / NaN): the plain chain for the rest
      float s = state_to_float(st);
      for (int i = start; i < kBlk; ++i) s = s + readlane_f(a, i);
      state_from_float(st, s);
      if (stats && lane == 0) atomicAdd(&stats->plain_blocks, 1u);
      break;
    }
    if (st.e == kNoInt) {
      const float s = st.s + readlane_f(a, start);
      state_from_float(st, s);
      ++start;
      ++n_float;
      ++barren;
      continue;
    }
    bool valid, up, tie;
    int I;
    decompose(a, st.e - 23, valid, I, up, tie);
    const bool act = lane >= start;
    const int q = lane_q(act && valid, I, up, tie, st.S & 1, nullptr);
    const int Sj = st.S + wave_incl_scan(q);
    const bool in = st.S > 0 ? (Sj >= kLoS && Sj <= kHiS) : (Sj <= -kLoS && Sj >= -kHiS);
    const unsigned long long bad = __builtin_amdgcn_ballot_w64(act && !(valid && in));
    const int first = bad ? __ffsll((long long)bad) - 1 : kBlk;
    if (first > start) {
      st.S = __builtin_amdgcn_readlane(Sj, first - 1);
      barren = 0;
    } else {
      ++barren;
    }
    if (first < kBlk) {
      const float s = to_float(st.S, st.e) + readlane_f(a, first);
      state_from_float(st, s);
      ++n_float;
    }
    start = first + 1;
  }
  if (stats && lane == 0) {
    atomicAdd(&stats->slow_blocks, 1u);
    atomicAdd(&stats->float_steps, n_float);
  }
}

// grid (n_spans, n_chains), ONE wave per workgroup.  out[span * out_stride + chain] = the sequential sum.
// alive (may be null): spans whose alive[span * alive_stride] is 0 are skipped (sum 0).
__global__ void __launch_bounds__(64) k_xs_chain(const Span* spans, const float* terms, size_t chain_stride,
                                                 const RecA* recA, const int* recB, size_t blk_stride, const RecA* topA,
                                                 const int* topB, size_t top_stride, float* out, int out_stride,
                                                 const int* alive, int alive_stride, Stats* stats)
{
  if (stats) stats += blockIdx.y & 7;  // (instrumentation: per chain, eight slots)
  // the terms of a super block whose record did not apply.  Such super blocks come in runs (a sum that hovers
  // around a power of two), so while one is evaluated the next one's records and terms are already on their
  // way into the other buffer: one exposed round trip to memory per run instead of one per super block.
  __shared__ float s_t[2][kSuper];
  const Span sp = spans[blockIdx.x];
  const int lane = threadIdx.x;
  if (alive && !alive[(size_t)blockIdx.x * alive_stride]) {
    if (lane == 0) out[(size_t)blockIdx.x * out_stride + blockIdx.y] = 0.f;
    return;
  }
  const float* t = terms + blockIdx.y * chain_stride + (size_t)sp.sb_off * kSuper;
  const RecA* ra = recA + blockIdx.y * blk_stride + (size_t)sp.sb_off * kSub;
  const int* rb = recB + blockIdx.y * blk_stride + (size_t)sp.sb_off * kSub;
  const RecA* ta = topA + blockIdx.y * top_stride + sp.sb_off;
  const int* tb = topB + blockIdx.y * top_stride + sp.sb_off;
  State st;
  st.e = kNoInt;
  st.S = 0;
  st.s = 0.f;
  unsigned hits = 0;
  unsigned have = 0xffffffffu;  // super block whose data the registers below hold (fetched, maybe still in flight)
  RecA br;
  int br_hi = 0;
  float tv[kSub];
  br.e = kInvalid;
  br.q0 = br.q1 = br.lo = 0;
#pragma unroll
  for (int k = 0; k < kSub; ++k) tv[k] = 0.f;
  auto fetch = [&](unsigned sbi) {
    const size_t b0 = (size_t)sbi * kSub;
    if (lane < kSub) {
      br = ra[b0 + lane];
      br_hi = rb[b0 + lane];
    }
#pragma unroll
    for (int k = 0; k < kSub; ++k) tv[k] = t[(b0 + k) * kBlk + lane];
    have = sbi;
  };
  for (unsigned t0 = 0; t0 < sp.n_sb; t0 += 64) {
    const unsigned n_here = min(64u, sp.n_sb - t0);
    RecA mine;
    int mine_hi = 0;
    mine.e = kInvalid;
    mine.q0 = mine.q1 = mine.lo = 0;
    if ((unsigned)lane < n_here) {
      mine = ta[t0 + lane];
      mine_hi = tb[t0 + lane];
    }
    for (unsigned i = 0; i < n_here; ++i) {
      if (apply(st, __builtin_amdgcn_readlane(mine.e, i), __builtin_amdgcn_readlane(mine.q0, i),
                __builtin_amdgcn_readlane(mine.q1, i), __builtin_amdgcn_readlane(mine.lo, i),
                __builtin_amdgcn_readlane(mine_hi, i))) {
        ++hits;
        continue;
      }
      // the super block's sixteen blocks one by one
      const unsigned sbi = t0 + i;
      if (have != sbi) fetch(sbi);
      float* buf = s_t[sbi & 1];
      const RecA cr = br;
      const int cr_hi = br_hi;
      __syncthreads();  // (one wave: orders the LDS accesses for the compiler, costs nothing)
#pragma unroll
      for (int k = 0; k < kSub; ++k) buf[k * kBlk + lane] = tv[k];
      __syncthreads();
      if (sbi + 1 < sp.n_sb) fetch(sbi + 1);  // lands while this one is evaluated
      for (int k = 0; k < kSub; ++k) {
        if (apply(st, __builtin_amdgcn_readlane(cr.e, k), __builtin_amdgcn_readlane(cr.q0, k),
                  __builtin_amdgcn_readlane(cr.q1, k), __builtin_amdgcn_readlane(cr.lo, k),
                  __builtin_amdgcn_readlane(cr_hi, k)))
          continue;
        slow_block(st, buf[k * kBlk + lane], stats);
      }
    }
  }
  if (lane == 0) {
    out[(size_t)blockIdx.x * out_stride + blockIdx.y] = state_to_float(st);
    if (stats) atomicAdd(&stats->top_hits, hits);
  }
}

}  // namespace xs
