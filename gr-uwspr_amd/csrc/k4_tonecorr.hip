// K4 -- the fine-grid correlation sweep: for every (freq, time-lag, drift)
// hypothesis and every one of the 162 symbols, the non-coherent correlation of
// the 256-sample symbol window against the four 4-FSK tone phasors.
//
// Reference: sync_and_demodulate_impl::sync_and_demodulate hot loop 3,
// lib/sync_and_demodulate_impl.cc:167-212 (per-symbol frequency cc:170-183,
// phasor recurrence cc:186-199, correlation cc:200-211).
//
// Mapping: the (hypothesis, symbol) pairs are flattened, g = 162*h + i.  A lane
// owns one pair and T of its 4 tones (T = 1, 2 or 4; a wavefront covers 16*T
// pairs).  The lane derives its symbol frequency in binary64 exactly as
// cc:173/cc:179 do, takes cos/sin of the per-sample phase step in binary64
// (cc:188-189), then walks the 256 samples in order, advancing its tone phasors
// by the reference's binary32 rotation recurrence (cc:193-195) and accumulating
// inp/quad in the reference's operand order (cc:206-207).  Every accumulator
// sees the reference's exact sequence of binary32 operations (no FMA, no tree
// reduction), so p[] is bit-identical.  T trades latency for overhead: T=1 gives
// 4x the wavefronts and a 4x shorter serial chain (the 5..17-hypothesis stages of
// the refinement schedule are latency-bound), T=4 amortises the sample reads
// over all four tones (the 200-hypothesis sweep is throughput-bound).
//
// Samples reach the lanes through LDS: a wavefront's symbol windows are runs of
// 2 KB in HBM; per 16-sample chunk the wave loads them cooperatively (16 lanes x
// 8 B = one 128-B run per window) into a per-wave LDS image whose rows are
// padded to 144 B, so the per-lane column reads (ds_read_b128, two samples each) are bank-conflict
// free (lanes of one pair read the same address: broadcast).  The next chunk's
// global loads are in flight while the current chunk is computed.
//
// Work per pair-sample: 32 correlation + 24 phasor binary32 ops = 56 VALU ops;
// the kernel is FP32-VALU bound; HBM sees each frame about once
// (L2 / Infinity Cache serve the re-reads by the other hypotheses of the frame).
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "k4_common.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K4_WAVES = 4;
constexpr int K4_ROWDW = 36;  // dwords per staged row: 16 samples x 8 B + 16 B pad (16-byte aligned rows)

template <int T, bool FAST = false>
__global__ __launch_bounds__(64 * K4_WAVES) void k4_tonecorr(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_hyp *__restrict__ hyps,
    int H, float *__restrict__ p_out, const dev_grp *__restrict__ taken, int hyps_per_grp, int skip_pairs) {
  // skip_pairs: stage S2, two tries per slot -- the slots k4_dpair computes (k4_drift_pair) are left alone here.
  // taken (or null): hypothesis h belongs to group h / hyps_per_grp; groups that have their phasor table
  // (dev_grp::nvalid bits 16..23) were computed by another kernel (k4_lag0) -- left alone here, nothing stored
  constexpr int PPW = 16 * T;        // pairs per wavefront
  constexpr int LPP = 4 / T;         // lanes per pair
  constexpr int NLD = PPW / 4;       // cooperative loads per lane per chunk
  __shared__ __align__(16) float lds_all[K4_WAVES][PPW * K4_ROWDW];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];

  const long long total = (long long)H * UWSPR_NSYM;
  const long long g0 = ((long long)xcd_swizzle(blockIdx.x, gridDim.x) * K4_WAVES + wv) * PPW;
  if (g0 >= total) return;  // wave-uniform; no workgroup barrier is used below

  // A wave's pairs span at most two hypotheses (162 > 64 >= PPW).
  const int hA = (int)(g0 / UWSPR_NSYM);
  const int iA0 = (int)(g0 - (long long)hA * UWSPR_NSYM);
  const int sb = min(PPW, UWSPR_NSYM - iA0);  // pairs < sb belong to hA
  dev_hyp A = hyps[hA];
  dev_hyp Bh = hyps[min(hA + 1, H - 1)];
  const int hB = min(hA + 1, H - 1);
  const bool takenA = (taken && ((taken[hA / hyps_per_grp].nvalid >> 16) & 0xff) != 0) ||
                      (skip_pairs && k4_drift_pair(hyps[hA & ~1], hyps[hA | 1], nframes));
  const bool takenB = (taken && ((taken[hB / hyps_per_grp].nvalid >> 16) & 0xff) != 0) ||
                      (skip_pairs && k4_drift_pair(hyps[hB & ~1], hyps[hB | 1], nframes));
  const bool okA = A.frame >= 0 && A.frame < nframes && !takenA;
  const bool okB = (hA + 1 < H) && Bh.frame >= 0 && Bh.frame < nframes && !takenB;
  // hypotheses that are skipped still own LDS rows: point them at safe samples
  if (!okA) { A.frame = 0; A.lag = 1 - 256 * iA0; }
  if (!okB) { Bh.frame = 0; Bh.lag = 1; }

  const int pr = lane / LPP;           // this lane's pair within the wave
  const int tone0 = (lane % LPP) * T;  // first of its T tones
  const bool mineA = pr < sb;
  const int own_i = mineA ? iA0 + pr : pr - sb;
  const bool own_ok = mineA ? okA : (okB && (g0 + pr) < total);
  // nothing live in this wavefront (a noise-only stream leaves most candidates' later stages dead): zeros, done
  if (!okA && !(okB && sb < PPW)) {   // wave-uniform
    if (g0 + pr < total && !(mineA ? takenA : takenB)) {
      float *out = p_out + (g0 + pr) * 4 + tone0;
#pragma unroll
      for (int j = 0; j < T; j++) out[j] = 0.0f;
    }
    return;
  }
  const int own_nb = (mineA ? A.lag : Bh.lag) + 256 * own_i;  // first sample index
  const bool interior = __all((own_nb > 0) && (own_nb + 255 < np));

  // ---- per-symbol tone phasor steps (binary64 angle, cc:173-189) ----------
  float cd[T], sd[T];
  {
    const dev_hyp &hy = mineA ? A : Bh;
    float fp;
    if (hy.m_type == UWSPR_LINEAR) {
      fp = (float)((double)hy.f0 +
                   ((double)hy.drift / 2.0) * ((double)(float)own_i - 81.0) / 81.0);
    } else {
      fp = hy.f0 + hy.slmc;
    }
#pragma unroll
    for (int j = 0; j < T; j++) {
      // delta[] = {-1.5,-0.5,0.5,1.5} * (float)(375/256), cc:148 (exact in binary32)
      const float delta = ((float)(tone0 + j) - 1.5f) * 1.46484375f;
      const double ang = kTwoPiDt * (double)(fp + delta);
      double sn, cs;
      sincos(ang, &sn, &cs);
      cd[j] = (float)cs;
      sd[j] = (float)sn;
    }
  }

  // ---- cooperative loader geometry ---------------------------------------
  // load t (0..NLD-1) of a chunk: window seg = 4t + lane/16, sample kk = lane%16.
  // (16-byte loads were tried: sample addresses are only 8-byte aligned and
  // unaligned dwordx4 loads ran 1.15-1.9x slower on gfx950.)
  const int kk = lane & 15;
  const int segq = lane >> 4;
  const float2 *src[NLD];  // sample 0 (+kk) of that window, fast path
#pragma unroll
  for (int t = 0; t < NLD; t++) {
    const int seg = 4 * t + segq;
    const bool sA = seg < sb;
    const int fr = sA ? A.frame : Bh.frame;
    const int nb = sA ? A.lag + 256 * (iA0 + seg) : Bh.lag + 256 * (seg - sb);
    src[t] = frames + ((long long)fr * fstride + nb + kk);
  }

  float2 stage[NLD];
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int t = 0; t < NLD; t++) stage[t] = src[t][16 * c];
    } else {
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int seg = 4 * t + segq;
        const bool sA = seg < sb;
        const long long fb = (long long)(sA ? A.frame : Bh.frame) * fstride;
        const int n = (sA ? A.lag + 256 * (iA0 + seg) : Bh.lag + 256 * (seg - sb)) + 16 * c + kk;
        const bool inr = (n > 0) && (n < np);  // cc:205, sample 0 excluded
        float2 v = frames[fb + min(max(n, 0), np - 1)];
        // a skipped sample contributes nothing: x*c with x = 0 leaves inp/quad unchanged
        stage[t] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };

  float c[T], s[T], inp[T], quad[T];
#pragma unroll
  for (int j = 0; j < T; j++) { c[j] = 1.0f; s[j] = 0.0f; inp[j] = 0.0f; quad[j] = 0.0f; }

  load_chunk(0);
  for (int ch = 0; ch < 16; ch++) {
    wave_lds_fence();  // previous chunk's reads are done before rows are rewritten
#pragma unroll
    for (int t = 0; t < NLD; t++)
      *reinterpret_cast<float2 *>(&lds[(4 * t + segq) * K4_ROWDW + 2 * kk]) = stage[t];
    wave_lds_fence();
    if (ch < 15) load_chunk(ch + 1);
    // two samples per LDS read (ds_read_b128); the next pair is in flight during these two steps
    float4 vnext = *reinterpret_cast<const float4 *>(&lds[pr * K4_ROWDW]);
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const float4 v = vnext;
      if (k < 14) vnext = *reinterpret_cast<const float4 *>(&lds[pr * K4_ROWDW + 2 * (k + 2)]);
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const float xx = half ? v.z : v.x, xy = half ? v.w : v.y;
#pragma unroll
        for (int j = 0; j < T; j++) {
          k4_mac<FAST>(inp[j], quad[j], xx, xy, c[j], s[j]);   // cc:206-207, left to right
          k4_rot<FAST>(c[j], s[j], cd[j], sd[j]);              // cc:193-195
        }
      }
    }
  }

  if (g0 + pr < total && !(mineA ? takenA : takenB)) {
    float *out = p_out + (g0 + pr) * 4 + tone0;  // p[pair][tone]
#pragma unroll
    for (int j = 0; j < T; j++) {
      const float pj = ieee_sqrtf(inp[j] * inp[j] + quad[j] * quad[j]);  // cc:211
      out[j] = own_ok ? pj : 0.0f;
    }
  }
}

void launch_tonecorr(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int H,
                     float4 *p, const dev_grp *taken, int hyps_per_grp, bool skip_pairs) {
  if (H <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, H, true);
  const long long total = (long long)H * UWSPR_NSYM;
  // T = tones per lane: 1 for the schedule's few-hypothesis stages (4x the wavefronts, a 4x shorter serial chain),
  // 2 from a million pairs up (the sweeps: amortise the sample reads; T = 4 -- 2 waves/SIMD at 200+ VGPRs -- measured
  // no faster).  Option "k4_t" forces it.
  int T = c->opt[UWSPR_OPT_K4_T];
  if (T != 1 && T != 2 && T != 4) T = total >= 1024 * 1024 ? 2 : 1;
  const long long waves = (total + 16 * T - 1) / (16 * T);
  const unsigned blocks = (unsigned)((waves + K4_WAVES - 1) / K4_WAVES);
  const float2 *fr = (const float2 *)frames;
  float *po = (float *)p;
  dim3 blk(64 * K4_WAVES);
  if (hyps_per_grp < 1) hyps_per_grp = 1;
  const int sp = skip_pairs && (H % 2 == 0) ? 1 : 0;
  if (T == 1 && c->fast_now) launch_timed(c, ps, (k4_tonecorr<1, true>), dim3(blocks), blk, 0, fr, c->fstride, c->np, B, hyps, H, po, taken, hyps_per_grp, sp);
  else if (T == 1) launch_timed(c, ps, k4_tonecorr<1>, dim3(blocks), blk, 0, fr, c->fstride, c->np, B, hyps, H, po, taken, hyps_per_grp, sp);
  else if (T == 2) launch_timed(c, ps, k4_tonecorr<2>, dim3(blocks), blk, 0, fr, c->fstride, c->np, B, hyps, H, po, taken, hyps_per_grp, sp);
  else launch_timed(c, ps, k4_tonecorr<4>, dim3(blocks), blk, 0, fr, c->fstride, c->np, B, hyps, H, po, taken, hyps_per_grp, sp);
}

}  // namespace uwspr

namespace uwspr {

constexpr int K4G_WAVES = 2;   // wavefronts per workgroup of the ring form

// ---------------------------------------------------------------------------
// Ring form of a lag group (dev_grp: up to 8 hypotheses that differ only in their time lag and so share their
// tone phasors -- the reference itself caches the tone tables across lags via `fplast`), for groups whose lags
// are ASCENDING and evenly spaced by STEP samples (S3: shift1-32..+32 step 16; S5: the jiggered shifts,
// step 8).  The NL windows of a (group, symbol) pair overlap almost entirely, so
// the pair's samples a = n - (lag[0] + 256 i), 0 <= a < 256 + (NL-1) STEP, are
// brought into LDS ONCE, as a ring of M = Q+1 slots of 16 samples, instead of
// once per lag: 4 global loads and 4 LDS stores per lane and 16-sample chunk
// instead of 4 NL.  Lag l at step k of chunk c reads a = 16 c + k + STEP l, i.e.
// slot (c + q) mod M, column r with (q, r) = divmod(k + STEP l, 16) known at
// compile time; the M slot addresses rotate once per chunk.  Every accumulator sees the
// reference's operation sequence (cc:193-195, 206-207).
// Phasor tables (k5_fold_schedule.hip: ptab_build; dev_grp::nvalid bits 16..23 = 1 + table of the slot, 0 = none):
// when every live group of a wavefront has one, the lanes do not run the recurrence (six instruction slots per
// sample step) but read c[k], s[k] from a 16-step slice the wavefront fetches per chunk into LDS.
constexpr int K4_PT_STRIDE = 18;   // float2 per (group, tone) row of the slice: 16 steps + pad (16-B rows on distinct banks)

template <int NL, int STEP, bool FAST = false>
__global__ __launch_bounds__(64 * K4G_WAVES) void k4_ring(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_grp *__restrict__ grps,
    int G, float *__restrict__ p_out, const float2 *__restrict__ ptab, int gps) {
  constexpr int PPW = 16;
  constexpr int W = (NL - 1) * STEP;       // extra samples beyond the first lag's window
  constexpr int Q = (15 + W) / 16;         // furthest slot a chunk reaches ahead
  constexpr int M = Q + 1;                 // ring slots
  constexpr int NSLOT = 16 + Q;            // slots a pair needs in all
  constexpr int RS = 32 * M + 4;           // dwords per pair row: 16-byte aligned, rows of a lane group on distinct banks
  __shared__ __align__(16) float lds_all[K4G_WAVES][PPW * RS];
  __shared__ __align__(16) float2 ptl_all[K4G_WAVES][8 * K4_PT_STRIDE];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];
  float2 *ptl = ptl_all[wv];

  const long long total = (long long)G * UWSPR_NSYM;
  const unsigned lblock = xcd_swizzle(blockIdx.x, gridDim.x);
  const long long g0 = ((long long)lblock * K4G_WAVES + wv) * PPW;
  if (g0 >= total) return;  // wave-uniform

  const int gA = (int)(g0 / UWSPR_NSYM);
  const int iA0 = (int)(g0 - (long long)gA * UWSPR_NSYM);
  const int sb = min(PPW, UWSPR_NSYM - iA0);  // pairs < sb belong to group gA
  const dev_grp A = grps[gA];
  const dev_grp Bg = grps[min(gA + 1, G - 1)];
  const bool okA = A.frame >= 0 && A.frame < nframes;
  const bool okB = (gA + 1 < G) && Bg.frame >= 0 && Bg.frame < nframes;
  const int frA = okA ? A.frame : 0, frB = okB ? Bg.frame : 0;
  const int nvA = okA ? (A.nvalid & 0xff) : 0, nvB = okB ? (Bg.nvalid & 0xff) : 0;
  // phasor tables: the table walk only when every live group of the wavefront has one (wave-uniform)
  const int selA = okA ? (A.nvalid >> 16) & 0xff : 0, selB = okB ? (Bg.nvalid >> 16) & 0xff : 0;
  const bool liveB = sb < PPW && okB;
  const bool use_tab = ptab != nullptr && (okA || liveB) && (!okA || selA != 0) && (!liveB || selB != 0);
  const float2 *tabA = ptab, *tabB = ptab;
  if (use_tab) {
    if (selA) tabA = ptab + ((size_t)(gA / gps) * kPtabPerSlot + (selA - 1)) * kPtabFloat2;
    if (selB) tabB = ptab + ((size_t)((gA + 1) / gps) * kPtabPerSlot + (selB - 1)) * kPtabFloat2;
    if (!selA) tabA = tabB;     // a dead group's lanes read some valid table (their results are discarded)
    if (!selB) tabB = tabA;
  }
  // nvalid bit 8: lag slot 2 repeats the previous stage's winner, its metric is known and nobody
  // reads its p[] -- skipped when that holds for every live group of the wave (NL == 5 only)
  const bool knownA = !okA || (A.nvalid & 0x100) != 0, knownB = !okB || (Bg.nvalid & 0x100) != 0;
  // (NL == 6, the jiggered shifts: the middle group's lag slot 2 is try 0, known when it repeats the stage-4 winner)
  const bool skip_mid = (NL == 5 || NL == 6) && knownA && (knownB || sb >= PPW);
  // (NL == 6: the last of a slot's three jiggered-shift groups holds five tries -- its lag slot 5 belongs to nobody and is
  // left out when that holds for every live group of the wave)
#ifndef UWSPR_SKIP_LAST
#define UWSPR_SKIP_LAST 1
#endif
  const bool skip_last = UWSPR_SKIP_LAST && NL == 6 && !skip_mid && (okA || liveB) && (!okA || nvA <= 5) && (!liveB || nvB <= 5);
  // first lag of the two groups; skipped groups point at safe samples
  const int l0A = okA ? A.lag[0] : 1 - 256 * iA0;
  const int l0B = okB ? Bg.lag[0] : 1;

  const int pr = lane >> 2;
  const int tone = lane & 3;
  const bool mineA = pr < sb;
  const int own_i = mineA ? iA0 + pr : pr - sb;
  // nothing live in this wavefront (candidates that were not worth a try leave their S3..S5 groups dead: on a
  // noise-only stream that is most of them): zeros for the dead groups' hypotheses, done
  if (!okA && !liveB) {   // wave-uniform
    if (g0 + pr < total) {
      const dev_grp &gy = mineA ? A : Bg;
      for (int l = 0; l < (gy.nvalid & 0xff) && l < NL; l++)
        p_out[((long long)(gy.hyp_base + (int)((gy.hmap >> (4 * l)) & 15u)) * UWSPR_NSYM + own_i) * 4 + tone] = 0.0f;
    }
    return;
  }
  const int own_nb = (mineA ? l0A : l0B) + 256 * own_i;
  const bool interior = __all((own_nb > 0) && (own_nb + 255 + 16 * Q < np)  /* the loader fetches whole slots */);

  // ---- this lane's tone phasor step (binary64 angle, cc:173-189) ------------
  float cd = 1.0f, sd = 0.0f;
  if (!use_tab) {   // (with tables the steps are in them)
    const dev_grp &gy = mineA ? A : Bg;
    float fp;
    if (gy.m_type == UWSPR_LINEAR) {
      fp = (float)((double)gy.f0 +
                   ((double)gy.drift / 2.0) * ((double)(float)own_i - 81.0) / 81.0);
    } else {
      fp = gy.f0 + gy.slmc;
    }
    const float delta = ((float)tone - 1.5f) * 1.46484375f;
    double sn, cs;
    sincos(kTwoPiDt * (double)(fp + delta), &sn, &cs);
    cd = (float)cs;
    sd = (float)sn;
  }

  // ---- cooperative loader: load j of a slot = pair 4j + lane/16, sample lane%16
  const int kk = lane & 15;
  const int segq = lane >> 4;
  const float2 *src[4];   // sample a = kk of that pair's ring (fast path)
  long long fbase[4];
  int nfirst[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pj = 4 * j + segq;
    const bool sA = pj < sb;
    fbase[j] = (long long)(sA ? frA : frB) * fstride;
    nfirst[j] = (sA ? l0A + 256 * (iA0 + pj) : l0B + 256 * (pj - sb)) + kk;
    src[j] = frames + fbase[j] + nfirst[j];
  }
  float2 stage[4];
  auto load_slot = [&](int sl) {
    if (interior) {
#pragma unroll
      for (int j = 0; j < 4; j++) stage[j] = src[j][16 * sl];
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int n = nfirst[j] + 16 * sl;
        const bool inr = (n > 0) && (n < np);  // cc:205, sample 0 excluded
        const float2 v = frames[fbase[j] + min(max(n, 0), np - 1)];
        stage[j] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto store_slot = [&](int pos) {   // pos = ring position (slot mod M), wave-uniform
#pragma unroll
    for (int j = 0; j < 4; j++)
      *reinterpret_cast<float2 *>(&lds[(4 * j + segq) * RS + 2 * kk + 32 * pos]) = stage[j];
  };

  // prologue: slots 0..Q -- all their loads in flight together (one memory round trip, not Q+1)
  {
    float2 pro[Q + 1][4];
#pragma unroll
    for (int sl = 0; sl <= Q; sl++) {
      load_slot(sl);
#pragma unroll
      for (int j = 0; j < 4; j++) pro[sl][j] = stage[j];
    }
#pragma unroll
    for (int sl = 0; sl <= Q; sl++) {
#pragma unroll
      for (int j = 0; j < 4; j++) stage[j] = pro[sl][j];
      store_slot(sl);
    }
  }

  // ring positions of slots c..c+Q as this lane's row addresses
  int sa[M];   // dword offsets into `lds` (kept as integers so the reads stay ds_read)
#pragma unroll
  for (int q = 0; q < M; q++) sa[q] = pr * RS + 32 * q;

  float c = 1.0f, s = 0.0f;
  float inp[NL], quad[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) { inp[l] = 0.0f; quad[l] = 0.0f; }

  // phasor-table slice of a chunk: 2 groups x 4 tones x 16 steps, two 8-byte loads per lane
  float2 tstage[2];
  auto load_tab = [&](int c) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int e = lane + 64 * j, tn = (e >> 4) & 3, st = e & 15;
      tstage[j] = ((e >> 6) ? tabB : tabA)[tn * 256 + 16 * c + st];
    }
  };
  auto store_tab = [&]() {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int e = lane + 64 * j;
      ptl[(e >> 4) * K4_PT_STRIDE + (e & 15)] = tstage[j];
    }
  };
  const int trow = ((mineA ? 0 : 4) + tone) * K4_PT_STRIDE;   // this lane's row of the slice
  if (use_tab) { load_tab(0); store_tab(); }

  int wpos = 0;  // ring position that slot c + Q + 1 will overwrite (= position of slot c)
  auto walk = [&](auto skip_tag, auto tab_tag) {
    constexpr int SKIPL = decltype(skip_tag)::value;   // lag slot left out: 2 (known), NL - 1 (unused), -1 (none)
    constexpr bool TAB = decltype(tab_tag)::value;     // phasors from the table slice, no recurrence
    for (int ch = 0; ch < 16; ch++) {
      // in flight during the chunk's arithmetic (the last chunk re-fetches the last slot: no
      // branch here or after the arithmetic, or the compiler sinks the arithmetic past it)
      load_slot(min(ch + Q + 1, NSLOT - 1));
      if (TAB) load_tab(min(ch + 1, 15));
      wave_lds_fence();                      // the slots written so far are visible
      // Two samples per read: STEP is even, so sample k + STEP l of an even step k and its
      // successor sit in one 16-byte-aligned LDS word pair -- one ds_read_b128 per lag and two
      // steps (an LDS read costs the SIMD about as much issue time as five arithmetic
      // instructions, tools/ring_probe.hip).  The reads of steps k+2, k+3 are issued before
      // the arithmetic of steps k, k+1 and pinned there.
      static_assert(STEP % 2 == 0, "sample pairs must not straddle lags");
      float4 vc[NL], vn[NL];
#pragma unroll
      for (int l = 0; l < NL; l++) {
        if (l == SKIPL) { vc[l] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); vn[l] = vc[l]; continue; }
        vc[l] = *reinterpret_cast<const float4 *>(&lds[sa[(STEP * l) >> 4] + 2 * ((STEP * l) & 15)]);
        vn[l] = vc[l];
      }
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        if (k < 14) {
#pragma unroll
          for (int l = 0; l < NL; l++) {
            if (l == SKIPL) continue;
            const int o = k + 2 + STEP * l;
            vn[l] = *reinterpret_cast<const float4 *>(&lds[sa[o >> 4] + 2 * (o & 15)]);
          }
        }
        // (pinning these reads ahead with sched_barrier -- 86 VGPRs instead of ~145 -- was 7 % slower
        // for S3 and 1 % slower for S5 under three streams: left to the scheduler)
        float4 ph = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (TAB) ph = *reinterpret_cast<const float4 *>(&ptl[trow + k]);   // (c, s) of steps k and k + 1
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const float pc = TAB ? (half ? ph.z : ph.x) : c, psn = TAB ? (half ? ph.w : ph.y) : s;
#pragma unroll
          for (int l = 0; l < NL; l++) {
            if (l == SKIPL) continue;
            const float xx = half ? vc[l].z : vc[l].x, xy = half ? vc[l].w : vc[l].y;
            k4_mac<FAST>(inp[l], quad[l], xx, xy, pc, psn);   // cc:206-207
          }
          if (!TAB) k4_rot<FAST>(c, s, cd, sd);      // cc:193-195
        }
#pragma unroll
        for (int l = 0; l < NL; l++) vc[l] = vn[l];
      }
      // slot ch is finished with: its position takes slot ch + Q + 1, and the addresses rotate
      wave_lds_fence();
      store_slot(wpos);
      if (TAB) store_tab();
      wpos = (wpos + 1 == M) ? 0 : wpos + 1;
      const int first = sa[0];
#pragma unroll
      for (int q = 0; q + 1 < M; q++) sa[q] = sa[q + 1];
      sa[M - 1] = first;
    }
  };
  using sk_none = std::integral_constant<int, -1>;
  using sk_mid = std::integral_constant<int, 2>;
  using sk_last = std::integral_constant<int, NL == 6 ? NL - 1 : -1>;
  if (use_tab) {
    if (skip_mid) walk(sk_mid{}, std::true_type{});
    else if (skip_last) walk(sk_last{}, std::true_type{});
    else walk(sk_none{}, std::true_type{});
  } else {
    if (skip_mid) walk(sk_mid{}, std::false_type{});
    else if (skip_last) walk(sk_last{}, std::false_type{});
    else walk(sk_none{}, std::false_type{});
  }

  if (g0 + pr < total) {
    const int nv = mineA ? nvA : nvB;
    const int hb = mineA ? A.hyp_base : Bg.hyp_base;
    const uint32_t hm = mineA ? A.hmap : Bg.hmap;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l < nv && !(skip_mid && l == 2)) {
        const float pj = ieee_sqrtf(inp[l] * inp[l] + quad[l] * quad[l]);  // cc:211
        p_out[((long long)(hb + (int)((hm >> (4 * l)) & 15u)) * UWSPR_NSYM + own_i) * 4 + tone] = pj;
      }
    }
    // groups that are skipped produce zeros for their hypotheses
    const dev_grp &gy = mineA ? A : Bg;
    if (!(mineA ? okA : okB) && (gy.nvalid & 0xff) > 0)
      for (int l = 0; l < (gy.nvalid & 0xff) && l < NL; l++)
        p_out[((long long)(gy.hyp_base + (int)((gy.hmap >> (4 * l)) & 15u)) * UWSPR_NSYM + own_i) * 4 + tone] = 0.0f;
  }
}

// lags of every group must be lag[0] + l*step, l < nvalid (the schedule's S3 / S5 emitters)
void launch_tonecorr_ring(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int G,
                          int NL, int step, int64_t nhyps, float4 *p, int gps) {
  if (G <= 0) return;
  const bool r5 = NL == 5 && step == 16, r6 = NL == 6 && step == 8;
  if (!r5 && !r6) return;   // (the schedule emits exactly these two spacings)
  prof_scope ps(c, UWSPR_K_TONECORR, nhyps, true);
  const long long total = (long long)G * UWSPR_NSYM;
  const long long waves = (total + 15) / 16;
  const unsigned blocks = (unsigned)((waves + K4G_WAVES - 1) / K4G_WAVES);
  const float2 *fr = (const float2 *)frames;
  float *po = (float *)p;
  dim3 blk(64 * K4G_WAVES);
  const float2 *pt = c->use_ptab ? c->d_ptab : nullptr;
  if (gps < 1) gps = 1;
  if (r5 && c->fast_now) launch_timed(c, ps, (k4_ring<5, 16, true>), dim3(blocks), blk, 0, fr, c->fstride, c->np, B, grps, G, po, pt, gps);
  else if (r5) launch_timed(c, ps, (k4_ring<5, 16>), dim3(blocks), blk, 0, fr, c->fstride, c->np, B, grps, G, po, pt, gps);
  else launch_timed(c, ps, (k4_ring<6, 8>), dim3(blocks), blk, 0, fr, c->fstride, c->np, B, grps, G, po, pt, gps);
}


}  // namespace uwspr

// ---------------------------------------------------------------------------
// The schedule's S0 (k4_lag0) and S1 / S4 (k4_fpack) on packed rows: the (slot, symbol) pairs of the whole launch
// are flattened and a workgroup of four wavefronts -- one per TONE -- takes 64 consecutive ones, one per lane.
namespace uwspr {

#ifndef K4F_CHUNK
#define K4F_CHUNK 32
#endif

// S0 of the schedule (cc:409-415: five lags 64 samples apart at one frequency), sample-major, for the slots
// whose frequency does not depend on the symbol (their phasor table exists: dev_grp::nvalid bits 16..23; the
// others are left to the flat kernel).  The four computed lags of a symbol read windows that overlap by three quarters
// (lag q covers samples [64 q, 64 q + 256) of the row's 448), so the row is streamed ONCE in 14 chunks of 32
// samples and every staged sample feeds each lag whose window it lies in, against phasor step k = a - 64 q of
// the slot's table (LDS, broadcast: the tone is wavefront-uniform) -- no recurrence, no per-lag reload.  The
// fifth lag is the first one symbol later: a 163rd, virtual row per slot supplies it for
// symbol 161.  Rows are packed like k4_fpack's: 64 consecutive (slot, row) pairs per workgroup, four tone
// wavefronts, at most two slots per workgroup.  Every accumulator still sees the reference's operation
// sequence (cc:206-207); the table is the recurrence of cc:193-195, computed once per slot.
constexpr int K4L_ROWS_PER_SLOT = UWSPR_NSYM + 1;

template <int QMASK, bool FAST>
__device__ __forceinline__ void k4_lag0_chunk(const float *__restrict__ smprow, const float4 *__restrict__ tabrow,
                                              int a0, float (&inp)[4], float (&quad)[4]) {
  // a0 = 32 c: stream position of the chunk's first sample; lag q sees it as step k = a0 - 64 q
  // eight steps per trip of a rolled loop: fully unrolled the scheduler hoists a chunk's LDS reads above its
  // arithmetic (512 registers and scratch)
#pragma unroll 1
  for (int k0 = 0; k0 < 32; k0 += 8) {
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
      const int k = k0 + kk;
      const float4 x = *reinterpret_cast<const float4 *>(&smprow[2 * k]);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!((QMASK >> q) & 1)) continue;
        const float4 ph = tabrow[(a0 - 64 * q + k) >> 1];      // (c, s) of steps k and k + 1 of lag q
        k4_mac<FAST>(inp[q], quad[q], x.x, x.y, ph.x, ph.y);   // cc:206-207
        k4_mac<FAST>(inp[q], quad[q], x.z, x.w, ph.z, ph.w);
      }
    }
  }
}

template <bool FAST = false>
__global__ __launch_bounds__(256, 4) void k4_lag0(   // 4 wavefronts per SIMD: <= 128 VGPRs (left alone the scheduler hoists a
                                                      // whole chunk's LDS reads above its arithmetic: 512 registers + scratch)
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_grp *__restrict__ grps,
    int nslots, float *__restrict__ p_out, const float2 *__restrict__ ptab) {
  constexpr int ROWS = 64, CH = 32, ROWDW = 2 * CH + 4, SEGS = 256 / CH, NR = ROWS / SEGS, NCHUNK = 14;
  __shared__ __align__(16) float smp[ROWS * ROWDW];
  __shared__ __align__(16) float4 tab[2][4][128];   // [slot A/B][tone][step pair]: the slot's whole table

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int tone = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long total = (long long)nslots * K4L_ROWS_PER_SLOT;
  const long long g0 = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * ROWS;
  if (g0 >= total) return;  // workgroup-uniform

  const int slotA = (int)(g0 / K4L_ROWS_PER_SLOT);
  const int rA0 = (int)(g0 - (long long)slotA * K4L_ROWS_PER_SLOT);
  const int sb = min(ROWS, K4L_ROWS_PER_SLOT - rA0);      // rows < sb belong to slot A
  const bool hasB = sb < ROWS && slotA + 1 < nslots;
  const int slotB = hasB ? slotA + 1 : slotA;
  const dev_grp A = grps[slotA], Bg = grps[slotB];
  // a slot is taken here only if it is live AND has its table (else the flat kernel's launch has it)
  const int selA = (A.nvalid >> 16) & 0xff, selB = (Bg.nvalid >> 16) & 0xff;
  const bool doA = A.frame >= 0 && A.frame < nframes && selA != 0;
  const bool doB = hasB && Bg.frame >= 0 && Bg.frame < nframes && selB != 0;
  if (!doA && !doB) return;  // workgroup-uniform

  // ---- the slots' tables -> LDS (8 KB each: 16 bytes x 2 loads per thread and slot)
#pragma unroll
  for (int sl = 0; sl < 2; sl++) {
    if (!(sl ? doB : doA)) continue;
    const float4 *src = reinterpret_cast<const float4 *>(
        ptab + ((size_t)(sl ? slotB : slotA) * kPtabPerSlot + ((sl ? selB : selA) - 1)) * kPtabFloat2);
    float4 *dst = &tab[sl][0][0];
    dst[tid] = src[tid];
    dst[tid + 256] = src[tid + 256];
  }

  const int row = lane;
  const bool mineA = row < sb;
  const int own_r = mineA ? rA0 + row : row - sb;       // 0..161 real symbols, 162 the virtual row
  const bool own_do = mineA ? doA : doB;
  const bool valid = (g0 + row < total) && (mineA || hasB) && own_do;

  // ---- loader: round r of a chunk = row SEGS r + tid/CH, sample tid%CH ----
  const int kk = tid % CH, seg = tid / CH;
  const float2 *fbA = frames + (long long)(doA ? A.frame : 0) * fstride;
  const float2 *fbB = frames + (long long)(doB ? Bg.frame : 0) * fstride;
  const int nA0 = A.lag[0] + 256 * rA0, nB0 = Bg.lag[0];
  const bool interior = doA && (nA0 > 0) && (nA0 + 256 * (sb - 1) + 32 * NCHUNK < np) &&
                        (sb >= ROWS || (doB && (nB0 > 0) && (nB0 + 256 * (ROWS - sb - 1) + 32 * NCHUNK < np)));
  float2 stage[NR];
  int nrow[NR];
  bool rowB[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int rw = SEGS * r + seg;
    rowB[r] = rw >= sb;
    nrow[r] = (rowB[r] ? nB0 + 256 * (rw - sb) : nA0 + 256 * rw) + kk;
  }
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int r = 0; r < NR; r++) stage[r] = (rowB[r] ? fbB : fbA)[nrow[r] + CH * c];
    } else {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int n = nrow[r] + CH * c;
        const bool inr = (n > 0) && (n < np);      // cc:205, sample 0 excluded
        const float2 v = (rowB[r] ? fbB : fbA)[min(max(n, 0), np - 1)];
        stage[r] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int r = 0; r < NR; r++)
      *reinterpret_cast<float2 *>(&smp[(SEGS * r + seg) * ROWDW + 2 * kk]) = stage[r];
  };

  float inp[4], quad[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { inp[q] = 0.0f; quad[q] = 0.0f; }
  const float *smprow = &smp[row * ROWDW];
  const float4 *tabrow = &tab[mineA ? 0 : 1][tone][0];

  load_chunk(0);
  // lag q is inside its window for chunks 2 q .. 2 q + 7: seven phases of two chunks with a fixed set of lags
  auto phase = [&](auto mask_tag, int c0) {
    constexpr int QMASK = decltype(mask_tag)::value;
#pragma unroll 1
    for (int c = c0; c < c0 + 2; c++) {
      __syncthreads();              // the previous chunk has been read by everyone (first: the tables are in)
      store_chunk();
      __syncthreads();
      load_chunk(min(c + 1, NCHUNK - 1));   // in flight during the arithmetic
      k4_lag0_chunk<QMASK, FAST>(smprow, tabrow, 32 * c, inp, quad);
    }
  };
  phase(std::integral_constant<int, 0x1>{}, 0);
  phase(std::integral_constant<int, 0x3>{}, 2);
  phase(std::integral_constant<int, 0x7>{}, 4);
  phase(std::integral_constant<int, 0xf>{}, 6);
  phase(std::integral_constant<int, 0xe>{}, 8);
  phase(std::integral_constant<int, 0xc>{}, 10);
  phase(std::integral_constant<int, 0x8>{}, 12);

  if (valid) {
    const int hb = mineA ? A.hyp_base : Bg.hyp_base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float pj = ieee_sqrtf(inp[q] * inp[q] + quad[q] * quad[q]);   // cc:211
      if (own_r < UWSPR_NSYM) p_out[((long long)(hb + q) * UWSPR_NSYM + own_r) * 4 + tone] = pj;
      // the wrap: (first lag, symbol r) is also (fifth lag, symbol r - 1)
      if (q == 0 && own_r >= 1) p_out[((long long)(hb + 4) * UWSPR_NSYM + own_r - 1) * 4 + tone] = pj;
    }
  }
}

// ---- round 6 (review item 4: "take the barriers out of k4_lag0"): two experimental forms of the same walk, option
// k4_forms bits 1 and 2, A/B in profiles/r06_k4_lag0_ab.txt.
//   WSETS = 1: the staged rows DOUBLE-BUFFERED -- chunk c + 1 is stored into the other buffer right behind the arithmetic
//              of chunk c: ONE workgroup barrier per chunk instead of two (34.8 + 16 KB of LDS: three workgroups per CU,
//              which is all a 256-candidate launch brings: 2.55 per CU).
//   WSETS = 2: the same, and the four lags of a row on TWO wavefronts per tone (lags {0, 2} and {1, 3}: both sets are
//              busy 12 of the 14 chunks): 8 wavefronts per workgroup, twice the wavefronts per SIMD for the same
//              arithmetic (a wavefront issues at most one instruction per 4.5 cycles: a launch of 2.5 wavefronts per
//              SIMD cannot fill the issue slots by itself), at the price of every staged sample being read by two lanes.
// The wrap (lag 4 = lag 0 one symbol later) and the operation order of every accumulator are the packed form's.
template <int WSETS>
__global__ __launch_bounds__(256 * WSETS, WSETS == 1 ? 3 : 6) void k4_lag0x(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_grp *__restrict__ grps,
    int nslots, float *__restrict__ p_out, const float2 *__restrict__ ptab) {
  constexpr int NT = 256 * WSETS;
  constexpr int ROWS = 64, CH = 32, ROWDW = 2 * CH + 4, SEGS = NT / CH, NR = ROWS / SEGS, NCHUNK = 14;
  __shared__ __align__(16) float smp[2][ROWS * ROWDW];
  __shared__ __align__(16) float4 tab[2][4][128];   // [slot A/B][tone][step pair]: the slot's whole table

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tone = wv & 3;
  const int wset = wv >> 2;                          // WSETS == 2: 0 = lags {0, 2}, 1 = lags {1, 3}
  const long long total = (long long)nslots * K4L_ROWS_PER_SLOT;
  const long long g0 = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * ROWS;
  if (g0 >= total) return;  // workgroup-uniform

  const int slotA = (int)(g0 / K4L_ROWS_PER_SLOT);
  const int rA0 = (int)(g0 - (long long)slotA * K4L_ROWS_PER_SLOT);
  const int sb = min(ROWS, K4L_ROWS_PER_SLOT - rA0);      // rows < sb belong to slot A
  const bool hasB = sb < ROWS && slotA + 1 < nslots;
  const int slotB = hasB ? slotA + 1 : slotA;
  const dev_grp A = grps[slotA], Bg = grps[slotB];
  const int selA = (A.nvalid >> 16) & 0xff, selB = (Bg.nvalid >> 16) & 0xff;
  const bool doA = A.frame >= 0 && A.frame < nframes && selA != 0;
  const bool doB = hasB && Bg.frame >= 0 && Bg.frame < nframes && selB != 0;
  if (!doA && !doB) return;  // workgroup-uniform

#pragma unroll
  for (int sl = 0; sl < 2; sl++) {
    if (!(sl ? doB : doA)) continue;
    const float4 *src = reinterpret_cast<const float4 *>(
        ptab + ((size_t)(sl ? slotB : slotA) * kPtabPerSlot + ((sl ? selB : selA) - 1)) * kPtabFloat2);
    float4 *dst = &tab[sl][0][0];
    for (int e = tid; e < 512; e += NT) dst[e] = src[e];
  }

  const int row = lane;
  const bool mineA = row < sb;
  const int own_r = mineA ? rA0 + row : row - sb;       // 0..161 real symbols, 162 the virtual row
  const bool own_do = mineA ? doA : doB;
  const bool valid = (g0 + row < total) && (mineA || hasB) && own_do;

  const int kk = tid % CH, seg = tid / CH;
  const float2 *fbA = frames + (long long)(doA ? A.frame : 0) * fstride;
  const float2 *fbB = frames + (long long)(doB ? Bg.frame : 0) * fstride;
  const int nA0 = A.lag[0] + 256 * rA0, nB0 = Bg.lag[0];
  const bool interior = doA && (nA0 > 0) && (nA0 + 256 * (sb - 1) + 32 * NCHUNK < np) &&
                        (sb >= ROWS || (doB && (nB0 > 0) && (nB0 + 256 * (ROWS - sb - 1) + 32 * NCHUNK < np)));
  float2 stage[NR];
  int nrow[NR];
  bool rowB[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int rw = SEGS * r + seg;
    rowB[r] = rw >= sb;
    nrow[r] = (rowB[r] ? nB0 + 256 * (rw - sb) : nA0 + 256 * rw) + kk;
  }
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int r = 0; r < NR; r++) stage[r] = (rowB[r] ? fbB : fbA)[nrow[r] + CH * c];
    } else {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int n = nrow[r] + CH * c;
        const bool inr = (n > 0) && (n < np);      // cc:205, sample 0 excluded
        const float2 v = (rowB[r] ? fbB : fbA)[min(max(n, 0), np - 1)];
        stage[r] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < NR; r++)
      *reinterpret_cast<float2 *>(&smp[buf][(SEGS * r + seg) * ROWDW + 2 * kk]) = stage[r];
  };

  float inp[4], quad[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { inp[q] = 0.0f; quad[q] = 0.0f; }
  const float4 *tabrow = &tab[mineA ? 0 : 1][tone][0];

  // chunk 0 into buffer 0, chunk 1 on its way; from then on: arithmetic of chunk c from buffer c & 1, chunk c + 1 into the
  // other buffer (its last readers passed the barrier that ended chunk c - 1), chunk c + 2 requested, ONE barrier
  load_chunk(0);
  store_chunk(0);
  load_chunk(1);
  __syncthreads();
  auto phase = [&](auto mask_tag, int c0) {
    constexpr int QM = decltype(mask_tag)::value;
#pragma unroll 1
    for (int c = c0; c < c0 + 2; c++) {
      if (WSETS == 1) {
        k4_lag0_chunk<QM, false>(&smp[c & 1][row * ROWDW], tabrow, 32 * c, inp, quad);
      } else if (wset == 0) {            // wavefront-uniform
        if (QM & 0x5) k4_lag0_chunk<QM & 0x5, false>(&smp[c & 1][row * ROWDW], tabrow, 32 * c, inp, quad);
      } else {
        if (QM & 0xa) k4_lag0_chunk<QM & 0xa, false>(&smp[c & 1][row * ROWDW], tabrow, 32 * c, inp, quad);
      }
      store_chunk((c + 1) & 1);
      load_chunk(min(c + 2, NCHUNK - 1));
      __syncthreads();
    }
  };
  phase(std::integral_constant<int, 0x1>{}, 0);
  phase(std::integral_constant<int, 0x3>{}, 2);
  phase(std::integral_constant<int, 0x7>{}, 4);
  phase(std::integral_constant<int, 0xf>{}, 6);
  phase(std::integral_constant<int, 0xe>{}, 8);
  phase(std::integral_constant<int, 0xc>{}, 10);
  phase(std::integral_constant<int, 0x8>{}, 12);

  if (valid) {
    const int hb = mineA ? A.hyp_base : Bg.hyp_base;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (WSETS == 2 && (q & 1) != wset) continue;
      const float pj = ieee_sqrtf(inp[q] * inp[q] + quad[q] * quad[q]);   // cc:211
      if (own_r < UWSPR_NSYM) p_out[((long long)(hb + q) * UWSPR_NSYM + own_r) * 4 + tone] = pj;
      // the wrap: (first lag, symbol r) is also (fifth lag, symbol r - 1)
      if (q == 0 && own_r >= 1) p_out[((long long)(hb + 4) * UWSPR_NSYM + own_r - 1) * 4 + tone] = pj;
    }
  }
}

void launch_tonecorr_lag0(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int nslots,
                          int64_t nhyps, float4 *p) {
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, nhyps, true);
  const unsigned wgs = (unsigned)(((long long)nslots * K4L_ROWS_PER_SLOT + 63) / 64);
  const int form = c->fast_now ? 0 : (c->opt[UWSPR_OPT_K4_FORMS] >> 1) & 3;     // bit 1: double-buffered; bit 2: + lag pairs
  if (c->fast_now)
    launch_timed(c, ps, k4_lag0<true>, dim3(wgs), dim3(256), 0, (const float2 *)frames, c->fstride, c->np, B, grps,
                 nslots, (float *)p, (const float2 *)c->d_ptab);
  else if (form & 2)
    launch_timed(c, ps, k4_lag0x<2>, dim3(wgs), dim3(512), 0, (const float2 *)frames, c->fstride, c->np, B, grps,
                 nslots, (float *)p, (const float2 *)c->d_ptab);
  else if (form & 1)
    launch_timed(c, ps, k4_lag0x<1>, dim3(wgs), dim3(256), 0, (const float2 *)frames, c->fstride, c->np, B, grps,
                 nslots, (float *)p, (const float2 *)c->d_ptab);
  else
    launch_timed(c, ps, k4_lag0<false>, dim3(wgs), dim3(256), 0, (const float2 *)frames, c->fstride, c->np, B, grps,
                 nslots, (float *)p, (const float2 *)c->d_ptab);
}

// Frequency stage (S1 / S4: the NF hypotheses of a candidate slot share frame, lag and drift model and differ only
// in f0, cc:416-419, 449-452): every lane correlates its symbol window against all NF frequencies -- the window is
// staged and read once for NF hypotheses.  When the per-symbol frequency does not depend on the symbol (drift == 0
// or the nonlinear model: the usual case) the phasor sequence c[k], s[k] (cc:186-199) is the same for all symbols
// of a (frequency, tone) -- the reference caches it via `fplast` for the same reason -- so lanes 0..39 of wavefront
// 0 advance the recurrences (a recurrence step costs its wavefront six instruction slots however few lanes run it),
// publish 32 steps at a time in LDS, and the other lanes only multiply-accumulate.  A workgroup spans at most two
// slots (162 > 64): two sets of hypothesis parameters, two phasor-table sets in LDS, each lane reading its own
// slot's.  Every accumulator and every phasor sees the reference's operation sequence.  Fallbacks inside: a slot whose frequency depends on
// the symbol (drifting linear model) puts the whole workgroup on per-lane recurrences; the known middle frequency
// is left out only when every live slot of the workgroup has it marked.
#ifdef K4F_STAMPS   // diagnostic build only (tools/k4f_stamps.py): where a k4_fpack wavefront's cycles go
constexpr int K4F_STAMP_WAVES = 4096;
__device__ unsigned long long g_k4f_stamps[K4F_STAMP_WAVES * 10];
#define K4F_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define K4F_ACC(a, t1, t0) a += (t1) - (t0)
#else
#define K4F_T(v) do { } while (0)
#define K4F_ACC(a, t1, t0) do { } while (0)
#endif

template <int NF, int CH, bool FAST = false>
__global__ __launch_bounds__(256) void k4_fpack(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_hyp *__restrict__ hyps,
    int nslots, float *__restrict__ p_out) {
  constexpr int ROWS = 64;
  constexpr int ROWDW = 2 * CH + 4;       // dwords per staged row: CH samples x 8 B + 16 B pad (16-byte aligned)
  constexpr int NCH = 256 / CH;           // chunks per symbol
  constexpr int SEGS = 256 / CH;          // rows staged per loader round
  constexpr int NR = ROWS / SEGS;
  __shared__ __align__(16) float smp[ROWS * ROWDW];
  __shared__ __align__(16) float4 tab[2][4][CH / 2][NF];   // [slot A/B][tone]: (c, s) of steps 2j and 2j+1 per frequency

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int tone = __builtin_amdgcn_readfirstlane(tid >> 6);
  #ifdef K4F_STAMPS
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_bar1 = 0, st_gen = 0, st_bar2 = 0, st_ar = 0;
#endif
  const long long total = (long long)nslots * UWSPR_NSYM;
  const long long g0 = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * ROWS;
  if (g0 >= total) return;  // workgroup-uniform
  const unsigned prio_class = k4_prio_class();

  const int slotA = (int)(g0 / UWSPR_NSYM);
  const int iA0 = (int)(g0 - (long long)slotA * UWSPR_NSYM);
  const int sb = min(ROWS, UWSPR_NSYM - iA0);            // rows < sb belong to slot A
  const bool hasB = sb < ROWS && slotA + 1 < nslots;
  const int slotB = hasB ? slotA + 1 : slotA;
  const dev_hyp hA = hyps[(size_t)slotA * NF], hB = hyps[(size_t)slotB * NF];
  float fA[NF], fB[NF];
#pragma unroll
  for (int q = 0; q < NF; q++) { fA[q] = hyps[(size_t)slotA * NF + q].f0; fB[q] = hyps[(size_t)slotB * NF + q].f0; }
  const bool liveA = hA.frame >= 0 && hA.frame < nframes;
  const bool liveB = hasB && hB.frame >= 0 && hB.frame < nframes;
  const bool tabA = (hA.m_type != UWSPR_LINEAR) || (hA.drift == 0.0f);     // fp independent of the symbol
  const bool tabB = (hB.m_type != UWSPR_LINEAR) || (hB.drift == 0.0f);
  const bool knownA = hyps[(size_t)slotA * NF + NF / 2].frame <= -2, knownB = hyps[(size_t)slotB * NF + NF / 2].frame <= -2;
  // workgroup-uniform modes
  const bool tabled = (!liveA || tabA) && (!liveB || tabB);
  const bool skip_mid = (liveA || liveB) && (!liveA || knownA) && (!liveB || knownB);

  const int row = lane;
  const bool mineA = row < sb;
  const int own_i = mineA ? iA0 + row : row - sb;
  const int own_slot = mineA ? slotA : slotB;
  const bool own_live = mineA ? liveA : liveB;
  const bool valid = (g0 + row < total) && (mineA || hasB);
  // nothing live in this workgroup (slots that were not worth a try): zeros, done (workgroup-uniform, before any barrier)
  if (!liveA && !liveB) {
    if (valid) {
#pragma unroll
      for (int q = 0; q < NF; q++) p_out[(((long long)own_slot * NF + q) * UWSPR_NSYM + own_i) * 4 + tone] = 0.0f;
    }
    return;
  }
  const dev_hyp &ho = mineA ? hA : hB;
  const float delta = ((float)tone - 1.5f) * 1.46484375f;                    // cc:148

  // ---- loader: round r of a chunk = row SEGS r + tid/CH, sample tid%CH ----
  const int kk = tid % CH, seg = tid / CH;
  const float2 *fbA = frames + (long long)(liveA ? hA.frame : 0) * fstride;
  const float2 *fbB = frames + (long long)(liveB ? hB.frame : 0) * fstride;
  const int nA0 = hA.lag + 256 * iA0, nB0 = hB.lag;                          // first sample of the first row of each slot
  const bool interior = liveA && (nA0 > 0) && (nA0 + 256 * sb < np) &&
                        (sb >= ROWS || (liveB && (nB0 > 0) && (nB0 + 256 * (ROWS - sb) < np)));   // workgroup-uniform
  float2 stage[NR];
  int nrow[NR];
  bool rowB[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int rw = SEGS * r + seg;
    rowB[r] = rw >= sb;
    nrow[r] = (rowB[r] ? nB0 + 256 * (rw - sb) : nA0 + 256 * rw) + kk;
  }
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int r = 0; r < NR; r++) stage[r] = (rowB[r] ? fbB : fbA)[nrow[r] + CH * c];
    } else {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int n = nrow[r] + CH * c;
        const bool inr = (n > 0) && (n < np);      // cc:205, sample 0 excluded
        const float2 v = (rowB[r] ? fbB : fbA)[min(max(n, 0), np - 1)];
        stage[r] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int r = 0; r < NR; r++)
      *reinterpret_cast<float2 *>(&smp[(SEGS * r + seg) * ROWDW + 2 * kk]) = stage[r];
  };

  float inp[NF], quad[NF];
#pragma unroll
  for (int q = 0; q < NF; q++) { inp[q] = 0.0f; quad[q] = 0.0f; }

  if (tabled) {
    // the 2 x NF x 4 recurrences (slot, frequency, tone) run on lanes 0 .. 8 NF - 1 of wavefront 0
    float cq = 1.0f, sq = 0.0f, cdq = 1.0f, sdq = 0.0f;
    const int gs = lane >= 4 * NF ? 1 : 0, gl = lane - gs * 4 * NF;
    const int gq = gl % NF, gt = min(gl / NF, 3);
    const bool gen = (tone == 0) && (lane < 8 * NF) && (gs ? liveB : liveA);
    if (gen) {
      float fq = gs ? fB[0] : fA[0];
#pragma unroll
      for (int q = 1; q < NF; q++) fq = (gq == q) ? (gs ? fB[q] : fA[q]) : fq;
      const dev_hyp &hg = gs ? hB : hA;
      const float fp = (hg.m_type == UWSPR_LINEAR)
                           ? (float)((double)fq + ((double)hg.drift / 2.0) * ((double)(float)0 - 81.0) / 81.0)
                           : fq + hg.slmc;                                   // cc:173 / cc:179 (drift == 0)
      const float gdelta = ((float)gt - 1.5f) * 1.46484375f;                 // cc:148
      double sn, cs;
      sincos(kTwoPiDt * (double)(fp + gdelta), &sn, &cs);
      cdq = (float)cs;
      sdq = (float)sn;
    }
    const float4 *mytab = &tab[mineA ? 0 : 1][tone][0][0];
    load_chunk(0);
    auto walk = [&](auto skip_tag) {
      constexpr bool SKIP = decltype(skip_tag)::value;
      for (int ch = 0; ch < NCH; ch++) {
        K4F_T(t0);
        __syncthreads();              // the previous chunk has been read by everyone
        K4F_T(t1);
        store_chunk();
        if (gen) {
#pragma unroll
          for (int k = 0; k < CH; k++) {
            reinterpret_cast<float2 *>(&tab[gs][gt][k >> 1][gq])[k & 1] = make_float2(cq, sq);
            k4_rot<FAST>(cq, sq, cdq, sdq);         // cc:193-195
          }
        }
        K4F_T(t2);
        __syncthreads();
        K4F_T(t3);
        k4_rotate_priority((unsigned)ch, prio_class);
        load_chunk(min(ch + 1, NCH - 1));  // in flight during the arithmetic (no branch around it)
#pragma unroll
        for (int k = 0; k < CH; k += 2) {
          // two samples and two phasor steps per LDS read (ds_read_b128)
          const float4 x = *reinterpret_cast<const float4 *>(&smp[row * ROWDW + 2 * k]);
#pragma unroll
          for (int q = 0; q < NF; q++) {
            if (SKIP && q == NF / 2) continue;
            const float4 ph = mytab[(k >> 1) * NF + q];   // two addresses per wavefront (slot A / slot B rows)
            k4_mac<FAST>(inp[q], quad[q], x.x, x.y, ph.x, ph.y);      // cc:206-207, step k
            k4_mac<FAST>(inp[q], quad[q], x.z, x.w, ph.z, ph.w);      // step k + 1
          }
        }
        K4F_T(t4);
        K4F_ACC(st_bar1, t1, t0); K4F_ACC(st_gen, t2, t1); K4F_ACC(st_bar2, t3, t2); K4F_ACC(st_ar, t4, t3);
      }
    };
    if (skip_mid) walk(std::true_type{}); else walk(std::false_type{});
  } else {
    float c[NF], s[NF], cd[NF], sd[NF];
#pragma unroll
    for (int q = 0; q < NF; q++) {
      const float f0 = mineA ? fA[q] : fB[q];
      const float fp = (ho.m_type == UWSPR_LINEAR)
                           ? (float)((double)f0 + ((double)ho.drift / 2.0) * ((double)(float)own_i - 81.0) / 81.0)  // cc:173
                           : f0 + ho.slmc;                                                                         // cc:179
      double sn, cs;
      sincos(kTwoPiDt * (double)(fp + delta), &sn, &cs);
      cd[q] = (float)cs; sd[q] = (float)sn; c[q] = 1.0f; s[q] = 0.0f;
    }
    load_chunk(0);
    for (int ch = 0; ch < NCH; ch++) {
      __syncthreads();
      store_chunk();
      __syncthreads();
      load_chunk(min(ch + 1, NCH - 1));
#pragma unroll
      for (int k = 0; k < CH; k += 2) {
        const float4 x4 = *reinterpret_cast<const float4 *>(&smp[row * ROWDW + 2 * k]);
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const float xx = half ? x4.z : x4.x, xy = half ? x4.w : x4.y;
#pragma unroll
          for (int q = 0; q < NF; q++) {
            k4_mac<FAST>(inp[q], quad[q], xx, xy, c[q], s[q]);        // cc:206-207
            k4_rot<FAST>(c[q], s[q], cd[q], sd[q]);                  // cc:193-195
          }
        }
      }
    }
  }

#ifdef K4F_STAMPS
  {
    const unsigned gw = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + tone;
    if (lane == 0 && gw < (unsigned)K4F_STAMP_WAVES) {
      unsigned long long *o = &g_k4f_stamps[(size_t)gw * 10];
      o[0] = st_c0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_r0; o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = st_bar1; o[5] = st_gen; o[6] = st_bar2; o[7] = st_ar;
      o[8] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
      o[9] = tabled ? 1 : 0;
    }
  }
#endif
  if (valid) {
#pragma unroll
    for (int q = 0; q < NF; q++) {
      float *o = &p_out[(((long long)own_slot * NF + q) * UWSPR_NSYM + own_i) * 4 + tone];
      if (!own_live) *o = 0.0f;                                   // skipped slot: its hypotheses read as zeros
      else if (!(skip_mid && q == NF / 2)) *o = ieee_sqrtf(inp[q] * inp[q] + quad[q] * quad[q]);   // cc:211
    }
  }
}

// hyps: nslots x 5 records; the 5 of a slot must share frame, lag, drift and model (S1 / S4)
void launch_tonecorr_fstage(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int nslots,
                            int64_t nhyps, float4 *p) {
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, nhyps, true);
  const unsigned wgs = (unsigned)(((long long)nslots * UWSPR_NSYM + 63) / 64);
  if (c->fast_now)
    launch_timed(c, ps, (k4_fpack<5, K4F_CHUNK, true>), dim3(wgs), dim3(256), 0,
                 (const float2 *)frames, c->fstride, c->np, B, hyps, nslots, (float *)p);
  else
    launch_timed(c, ps, (k4_fpack<5, K4F_CHUNK>), dim3(wgs), dim3(256), 0,
                 (const float2 *)frames, c->fstride, c->np, B, hyps, nslots, (float *)p);
}

}  // namespace uwspr

#ifdef K4F_STAMPS
extern "C" int uwspr_debug_k4f_stamps(unsigned long long *out, int nwaves) {
  if (nwaves > uwspr::K4F_STAMP_WAVES) nwaves = uwspr::K4F_STAMP_WAVES;
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(uwspr::g_k4f_stamps), (size_t)nwaves * 80);
}
#endif
