// k4_pair.hip -- K4 for the schedule's stage S2 (the two drift tries, sync_and_demodulate_impl.cc:423-433):
// one phasor recurrence serves both tries.
//
// S2 evaluates (f1, shift1) at drift1 + 0.5 and drift1 - 0.5 (cc:425, 429).  With a linear model the tone frequency
// of symbol i is  fp = (float)((double)f0 + ((double)drift / 2.0) * ((double)(float)i - 81.0) / 81.0)  (cc:173), so
// every symbol of a try has its own phasor sequence c[k], s[k] (cc:186-196) and the flat kernel spends 6 of its 14
// instructions per sample on that recurrence.  When the candidate came in WITHOUT drift -- every candidate of an FDR
// with maxdrift = 0 -- the tries are +d and -d, and
//     (+d / 2) * (i - 81) / 81   ==   (-d / 2) * ((162 - i) - 81) / 81      bit for bit
// (the same two magnitudes multiplied, the same quotient; IEEE sign symmetry), hence fp and with it the whole
// sequence of symbol i under +d IS the sequence of symbol 162 - i under -d.  A lane here owns a pair-row
// r = 0..162 of a slot and one tone: it runs the recurrence once and correlates two windows against it -- symbol r of
// the + try and symbol 162 - r of the - try (r = 162 and r = 0 have only one of the two) -- 22 instructions per
// sample for two terms instead of 28, and half the binary64 sincos.  Every accumulator still sees the reference's
// operations in the reference's order (cc:193-195, 206-207): results are byte-identical to the flat kernel's
// (tests/test_gpu_parity.py: test_schedule_forms_are_identical, test_drift_pairs_*).
//
// Slots whose two tries do not mirror each other (a candidate with drift, lazy / dead slots) are left to the flat
// kernel (k4_tonecorr with skip_pairs = 1 uses the same predicate).  Layout and loader are the flat kernel's:
// 16 pair-rows x 4 tones per wavefront, 16-sample chunks through a per-wavefront LDS image (32 rows here).
#include "k4_common.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K4P_WAVES = 4;
constexpr int K4P_ROWDW = 36;    // dwords per staged row: 16 samples x 8 B + 16 B pad
constexpr int K4P_PAIRS = UWSPR_NSYM + 1;   // pair-rows per slot: r = 0..162

template <bool FAST>
__global__ __launch_bounds__(64 * K4P_WAVES) void k4_dpair(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_hyp *__restrict__ hyps,
    int nslots, float *__restrict__ p_out) {
  constexpr int PPW = 16;        // pair-rows per wavefront
  constexpr int NLD = 8;         // cooperative loads per lane and chunk: 32 windows x 16 samples / 64 lanes
  __shared__ __align__(16) float lds_all[K4P_WAVES][2 * PPW * K4P_ROWDW];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];

  const long long total = (long long)nslots * K4P_PAIRS;
  const long long g0 = ((long long)xcd_swizzle(blockIdx.x, gridDim.x) * K4P_WAVES + wv) * PPW;
  if (g0 >= total) return;   // wave-uniform; no workgroup barrier below

  // a wavefront's pair-rows span at most two slots (163 > 16)
  const int sA = (int)(g0 / K4P_PAIRS);
  const int rA0 = (int)(g0 - (long long)sA * K4P_PAIRS);
  const int sb = min(PPW, K4P_PAIRS - rA0);          // pair-rows < sb belong to slot sA
  const int sB = min(sA + 1, nslots - 1);
  dev_hyp Ap = hyps[2 * sA], Am = hyps[2 * sA + 1], Bp = hyps[2 * sB], Bm = hyps[2 * sB + 1];
  const bool hasB = (sA + 1 < nslots) && sb < PPW;
  const bool okA = k4_drift_pair(Ap, Am, nframes);
  const bool okB = hasB && k4_drift_pair(Bp, Bm, nframes);
  // slots without a live try get zeros, as from the flat kernel (which may not run at all: uwspr_api.hip)
  const bool deadA = Ap.frame < 0 && Am.frame < 0, deadB = hasB && Bp.frame < 0 && Bm.frame < 0;

  const int pr = lane >> 2;            // this lane's pair-row within the wavefront
  const int tone = lane & 3;
  const bool mineA = pr < sb;
  const int own_r = mineA ? rA0 + pr : pr - sb;
  const bool in_range = (g0 + pr) < total;
  const bool own_ok = mineA ? okA : (okB && in_range);
  const bool own_dead = in_range && (mineA ? deadA : deadB);
  const int own_slot = mineA ? sA : sB;
  float *outP = p_out + ((long long)(2 * own_slot) * UWSPR_NSYM + min(own_r, UWSPR_NSYM - 1)) * 4 + tone;
  float *outM = p_out + ((long long)(2 * own_slot + 1) * UWSPR_NSYM + min(UWSPR_NSYM - own_r, UWSPR_NSYM - 1)) * 4 + tone;
  if (!okA && !okB) {        // wave-uniform: nothing of this wavefront is a mirrored pair
    if (own_dead) {
      if (own_r < UWSPR_NSYM) *outP = 0.0f;
      if (own_r >= 1) *outM = 0.0f;
    }
    return;
  }
  const dev_hyp &hy = mineA ? Ap : Bp;

  // ---- the pair's tone phasor step (binary64 angle, cc:173-189), from the + try at pair-row r ----
  float cd, sd;
  {
    const float fp = (float)((double)hy.f0 + ((double)hy.drift / 2.0) * ((double)(float)own_r - 81.0) / 81.0);   // cc:173
    k4_tone_step(fp, tone, cd, sd);
  }

  // ---- cooperative loader: load t of a chunk fills LDS row 4 t + lane / 16, sample lane % 16;
  //      rows 0..15 = the P windows of the 16 pair-rows, rows 16..31 = their M windows ----
  const int kk = lane & 15;
  const int segq = lane >> 4;
  // window P of pair-row r: symbol r of the + try (r <= 161); window M: symbol 162 - r of the - try (r >= 1); the two
  // tries share frame and lag (k4_drift_pair).  A missing window, and every window of a slot that is not ours, reads
  // safe samples of frame 0 instead and is not stored.
  int nb[NLD];         // first sample index of the window
  long long fb[NLD];   // frame base
  bool inside = true;
#pragma unroll
  for (int t = 0; t < NLD; t++) {
    const int row = 4 * t + segq;
    const int q = row & 15;
    const bool isM = row >= 16;
    const bool qA = q < sb;
    const int r = qA ? rA0 + q : q - sb;
    const int sym = isM ? UWSPR_NSYM - r : r;
    const dev_hyp &h = qA ? Ap : Bp;
    const bool real = (qA ? okA : okB) && sym >= 0 && sym < UWSPR_NSYM;
    nb[t] = real ? h.lag + 256 * sym : 1 + 256 * q;
    fb[t] = real ? (long long)h.frame * fstride : 0;
    inside = inside && (nb[t] > 0) && (nb[t] + 255 < np);
  }
  const bool interior = __all(inside);

  float2 stage[NLD];
  auto load_chunk = [&](int c) {
    if (interior) {
#pragma unroll
      for (int t = 0; t < NLD; t++) stage[t] = frames[fb[t] + nb[t] + 16 * c + kk];
    } else {
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int n = nb[t] + 16 * c + kk;
        const bool inr = (n > 0) && (n < np);   // cc:205, sample 0 excluded
        const float2 v = frames[fb[t] + min(max(n, 0), np - 1)];
        stage[t] = inr ? v : make_float2(0.0f, 0.0f);   // a skipped sample contributes nothing
      }
    }
  };

  float c = 1.0f, s = 0.0f, inpP = 0.0f, quadP = 0.0f, inpM = 0.0f, quadM = 0.0f;
  const float *rowP = &lds[pr * K4P_ROWDW], *rowM = &lds[(16 + pr) * K4P_ROWDW];

  load_chunk(0);
  for (int ch = 0; ch < 16; ch++) {
    wave_lds_fence();   // the previous chunk's reads are done before the rows are rewritten
#pragma unroll
    for (int t = 0; t < NLD; t++)
      *reinterpret_cast<float2 *>(&lds[(4 * t + segq) * K4P_ROWDW + 2 * kk]) = stage[t];
    wave_lds_fence();
    if (ch < 15) load_chunk(ch + 1);
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const float4 vp = *reinterpret_cast<const float4 *>(&rowP[2 * k]);   // two samples per LDS read
      const float4 vm = *reinterpret_cast<const float4 *>(&rowM[2 * k]);
      k4_mac<FAST>(inpP, quadP, vp.x, vp.y, c, s);    // cc:206-207, step k, + try
      k4_mac<FAST>(inpM, quadM, vm.x, vm.y, c, s);    //                      - try: the same phasor
      k4_rot<FAST>(c, s, cd, sd);                     // cc:193-195
      k4_mac<FAST>(inpP, quadP, vp.z, vp.w, c, s);    // step k + 1
      k4_mac<FAST>(inpM, quadM, vm.z, vm.w, c, s);
      k4_rot<FAST>(c, s, cd, sd);
    }
  }

  if (own_ok) {
    if (own_r < UWSPR_NSYM) *outP = ieee_sqrtf(inpP * inpP + quadP * quadP);   // cc:211
    if (own_r >= 1) *outM = ieee_sqrtf(inpM * inpM + quadM * quadM);
  } else if (own_dead) {
    if (own_r < UWSPR_NSYM) *outP = 0.0f;
    if (own_r >= 1) *outM = 0.0f;
  }
}

void launch_tonecorr_dpair(uwspr_ctx *c, const float *frames, int B, const dev_hyp *hyps, int nslots, float4 *p) {
  if (nslots <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, 2 * nslots, true);
  const long long waves = ((long long)nslots * K4P_PAIRS + 15) / 16;
  const unsigned blocks = (unsigned)((waves + K4P_WAVES - 1) / K4P_WAVES);
  if (c->fast_now)
    launch_timed(c, ps, k4_dpair<true>, dim3(blocks), dim3(64 * K4P_WAVES), 0, (const float2 *)frames, c->fstride, c->np, B,
                 hyps, nslots, (float *)p);
  else
    launch_timed(c, ps, k4_dpair<false>, dim3(blocks), dim3(64 * K4P_WAVES), 0, (const float2 *)frames, c->fstride, c->np, B,
                 hyps, nslots, (float *)p);
}

}  // namespace uwspr
