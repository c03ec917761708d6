// fft512_lane.h -- per-lane arithmetic of the 512-point spectrogram FFT (K1).
//
// One 64-lane wavefront transforms one 512-sample row; every lane owns 8
// complex points and runs three register-resident radix-8 passes, each of
// which is three fused radix-2 decimation-in-time stages.  The arithmetic is
// the textbook iterative radix-2 DIT FFT on bit-reversed input
//     t = W*x[b+j+h];  x[b+j] = u + t;  x[b+j+h] = u - t
// stage for stage and butterfly for butterfly, only re-associated across lanes,
// so results are bit-identical to a scalar radix-2 DIT using the same twiddle
// table (binary32, no FMA contraction).  The reference computes this with
// FFTW3f (lib/FDR_impl.cc:123-132,244), a third-party library whose rounding
// is unpinned.
//
// Index algebra (p = position in the bit-reversed array, 9 bits):
//   pass A (h=1,2,4):    lane L holds p = 8*rev6(L) + r,      r = 0..7
//   pass B (h=8,16,32):  lane L holds p = 64*(L>>3) + 8*b + (L&7), b = 0..7
//   pass C (h=64..256):  lane L holds p = 64*a + L,           a = 0..7
// Functions are host+device so the lane algebra is unit-tested by emulating
// the 64 lanes on the CPU (tests/host/fft_lane_emul.cc, tests/test_host_blocks.py).
#pragma once

#ifdef __HIPCC__
#define UWSPR_HD __host__ __device__ __forceinline__
#else
#define UWSPR_HD inline
#endif

namespace uwspr {

struct cpx { float r, i; };

UWSPR_HD int rev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
UWSPR_HD int rev6(int x) { return (rev3(x & 7) << 3) | rev3((x >> 3) & 7); }

// LDS image index of position p used by both exchanges: an XOR swizzle of the low five bits by
// the high four, chosen so that all four access patterns are bank-conflict free on gfx950 (8-byte
// elements: ds_write_b64 is served in groups of 16 lanes over 32 banks -> index mod 16 must differ
// within a group; ds_read_b64 in groups of 32 lanes over 64 banks -> index mod 32):
//   write A  varies p8 p7 p6 p5          -> index bits 2 1 3 0
//   read  B  varies p7 p6 p2 p1 p0       -> index bits 4 3 (2 1 0 xor const)
//   write B  varies p6 p2 p1 p0          -> index bits 3 (2 1 0 xor const)
//   read  C  varies p4..p0               -> index bits 4..0 xor const
// (a one-per-32 padding left read B with up to four-way conflicts: the exchanges, not the
// arithmetic, bounded K1).
UWSPR_HD int xidx(int p) {
  return p ^ ((((p >> 7) & 1) << 4) | (((p >> 6) & 1) << 3) | (((p >> 8) & 1) << 2) | (((p >> 7) & 1) << 1) |
              ((p >> 5) & 1));
}
constexpr int XCHG_LEN = 512;

// u,v <- u + w*v, u - w*v   (general twiddle)
UWSPR_HD void bfly(cpx &u, cpx &v, float wr, float wi) {
  float tr = wr * v.r - wi * v.i;
  float ti = wr * v.i + wi * v.r;
  float ur = u.r, ui = u.i;
  u.r = ur + tr; u.i = ui + ti;
  v.r = ur - tr; v.i = ui - ti;
}
// w = 1
UWSPR_HD void bfly1(cpx &u, cpx &v) {
  float tr = v.r, ti = v.i, ur = u.r, ui = u.i;
  u.r = ur + tr; u.i = ui + ti;
  v.r = ur - tr; v.i = ui - ti;
}
// w = -i :  t = (v.i, -v.r)
UWSPR_HD void bflymi(cpx &u, cpx &v) {
  float tr = v.i, ti = -v.r, ur = u.r, ui = u.i;
  u.r = ur + tr; u.i = ui + ti;
  v.r = ur - tr; v.i = ui - ti;
}

// Twiddles one lane needs for one radix-8 pass: 1 for the first fused stage,
// 2 for the second, 4 for the third.
struct pass_tw { cpx s1, s2[2], s3[4]; };

// tw = table [256] of (cos, -sin)(2*pi*k/512).  unit = the lane's j offset
// (r for pass B, c for pass C); hs = first half-size of the pass (8 or 64).
UWSPR_HD pass_tw load_pass_tw(const cpx *tw, int unit, int hs) {
  pass_tw t;
  t.s1 = tw[unit * (256 / hs)];
#pragma unroll
  for (int e = 0; e < 2; e++) t.s2[e] = tw[(hs * e + unit) * (256 / (2 * hs))];
#pragma unroll
  for (int e = 0; e < 4; e++) t.s3[e] = tw[(hs * e + unit) * (256 / (4 * hs))];
  return t;
}

// Pass A: registers y[r] hold positions 8q+r.  Twiddles are the constants
// tw[0]=1, tw[128]=-i, tw[64], tw[192].
UWSPR_HD void pass_a(cpx y[8], cpx w64, cpx w192) {
  bfly1(y[0], y[1]); bfly1(y[2], y[3]); bfly1(y[4], y[5]); bfly1(y[6], y[7]);
  bfly1(y[0], y[2]); bflymi(y[1], y[3]); bfly1(y[4], y[6]); bflymi(y[5], y[7]);
  bfly1(y[0], y[4]);
  bfly(y[1], y[5], w64.r, w64.i);
  bflymi(y[2], y[6]);
  bfly(y[3], y[7], w192.r, w192.i);
}

// Pass B / C: registers v[e], e = 0..7 = the 3 bits being combined.
UWSPR_HD void pass_bc(cpx v[8], const pass_tw &t) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) bfly(v[e], v[e + 1], t.s1.r, t.s1.i);
#pragma unroll
  for (int g = 0; g < 8; g += 4) {
    bfly(v[g + 0], v[g + 2], t.s2[0].r, t.s2[0].i);
    bfly(v[g + 1], v[g + 3], t.s2[1].r, t.s2[1].i);
  }
#pragma unroll
  for (int e = 0; e < 4; e++) bfly(v[e], v[e + 4], t.s3[e].r, t.s3[e].i);
}

// Pass C when only bins 0..63 and 448..511 are wanted (slots 0 and 7: a pass band within 64 columns of
// the centre, which every shipped flowgraph has).  The wanted outputs are computed by exactly the
// butterflies of pass_bc -- same operands, same operations -- and the others are left out: the
// first fused stage in full, then one output of each second-stage butterfly (slots 0, 4 from the
// sums, 3, 7 from the differences) and one output of two third-stage butterflies.  88 operations
// instead of 120.  Slots 1..6 of v are undefined afterwards.
UWSPR_HD void bfly_sum(cpx &u, const cpx &v, float wr, float wi) {   // u <- u + w*v
  float tr = wr * v.r - wi * v.i;
  float ti = wr * v.i + wi * v.r;
  u.r = u.r + tr; u.i = u.i + ti;
}
UWSPR_HD void bfly_diff(const cpx &u, cpx &v, float wr, float wi) {  // v <- u - w*v
  float tr = wr * v.r - wi * v.i;
  float ti = wr * v.i + wi * v.r;
  v.r = u.r - tr; v.i = u.i - ti;
}
UWSPR_HD void pass_c_narrow(cpx v[8], const pass_tw &t) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) bfly(v[e], v[e + 1], t.s1.r, t.s1.i);
  bfly_sum(v[0], v[2], t.s2[0].r, t.s2[0].i);
  bfly_sum(v[4], v[6], t.s2[0].r, t.s2[0].i);
  bfly_diff(v[1], v[3], t.s2[1].r, t.s2[1].i);
  bfly_diff(v[5], v[7], t.s2[1].r, t.s2[1].i);
  bfly_sum(v[0], v[4], t.s3[0].r, t.s3[0].i);
  bfly_diff(v[3], v[7], t.s3[3].r, t.s3[3].i);
}

// Which input sample (0..511) lane L keeps in register slot r before pass A:
// position p = 8*rev6(L)+r holds x[rev9(p)] = x[L + 64*rev3(r)].
UWSPR_HD int in_sample(int L, int r) { return L + 64 * rev3(r); }
// exchange 1: lane L writes slot r to position...
UWSPR_HD int posA(int L, int r) { return 8 * rev6(L) + r; }
// ...and lane L reads slot b for pass B from position
UWSPR_HD int posB(int L, int b) { return 64 * (L >> 3) + 8 * b + (L & 7); }
// exchange 2: lane L reads slot a for pass C from position
UWSPR_HD int posC(int L, int a) { return 64 * a + L; }
// after pass C lane L slot a holds DFT bin k = 64a+L; fftshifted column
// (FDR_impl.cc:247-248: ps[i][j] = |X[(j+spb) mod size]|^2) is j = k ^ 256.
UWSPR_HD int out_col(int L, int a) { return (64 * a + L) ^ 256; }

}  // namespace uwspr
