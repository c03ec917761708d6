// K4, stage S5 of the schedule (sync_and_demodulate_impl.cc:457-468: the 17 jiggered shifts shift1 + 8 ii at the
// refined frequency, mode 2) with the sample pairs of a lag group in a REGISTER ring.
//
// A lag group (dev_grp: six / six / five of a slot's tries, lags ascending, 8 samples apart) shares its tone phasors:
// lag l at steps 2j, 2j + 1 multiplies phasor pair j with sample pair j + 4 l.  k4_ring<6,8> (k4_tonecorr.hip) keeps the
// group's window in LDS and fetches that sample pair for every lag -- six ds_read_b128 per iteration, each writing four
// VGPRs through the port the arithmetic writes its results through (DESIGN: an LDS read of 16 bytes costs the SIMD about
// four arithmetic issue slots), 7 reads per 96 multiply-adds with the phasor pair.  (What it buys and what it does not:
// profiles/r05_k4_jig_ab.txt -- the wavefront's issue stalls go, its third neighbour on the SIMD goes too: 220 VGPRs.)  But a sample pair is the same pair for
// all six lags, at iterations 4 apart: here iteration j fetches ONE new pair (j + 21) into a ring of 24 float4 that the
// lags index at compile time (the walk is unrolled over the ring period: 24 iterations = three 16-step chunks): 2 LDS
// reads per 96 multiply-adds.  Same operands into the same accumulators in the same order as cc:206-207; the phasors
// are the slot's table (the recurrence of cc:193-195, built once: k5_fold_schedule.hip: ptab_build).
//
// Layout as in k4_ring: a wavefront = 16 (group, symbol) pairs x 4 tones, wavefront-private LDS (a ring of four 16-sample
// slots per pair, filled by the wavefront's own loader; a 16-step slice of the phasor tables per chunk), no workgroup
// barrier.  Wavefronts whose live groups do not all have a table (drifting linear candidates -- stage 2 can give ANY
// candidate a drift of +-0.5 Hz -- : the tone frequency depends on the symbol) run the same walk on per-lane phasor
// recurrences instead of the table slice.
// The lag slot that is known (try 0 = the stage-4 winner) or unused (the third group has five tries) is left out by
// instantiating the walk three times (12 KB of code each: the three run side by side on a CU and fit its instruction
// cache; three more for the recurrence walk, rarely run); a wavefront-uniform branch per lag instead costs the wavefront a dozen scalar instructions per 96 multiply-adds,
// and a wavefront issues ONE instruction of any kind per 4.5 cycles (measured: 125 -> 152 us).
#include "k4_common.h"

#pragma clang fp contract(off)

namespace uwspr {

constexpr int K4J_WAVES = 2;          // wavefronts per workgroup (independent)
#ifndef K4J_OCC
#define K4J_OCC 2
#endif
constexpr int K4J_PT_STRIDE = 18;     // float2 per (group, tone) row of the table slice: 16 steps + pad

__global__ __launch_bounds__(64 * K4J_WAVES, K4J_OCC) void k4_jig(
    const float2 *__restrict__ frames, int fstride, int np, int nframes, const dev_grp *__restrict__ grps,
    int G, float *__restrict__ p_out, const float2 *__restrict__ ptab, int gps) {
  constexpr int NL = 6, PPW = 16;
  constexpr int R = 24;                    // ring entries (sample pairs)
  constexpr int AH = 21;                   // iteration j fetches pair j + AH (one iteration before lag 5 reads it: 22 of
                                           // the 24 entries are live); its lags read pairs j, j + 4, .., j + 20
  constexpr int Q = (2 * (7 + AH) + 1) / 16;   // furthest slot a chunk's reads reach: sample 2 * 7 + 2 * AH + 1 = 61
  constexpr int M = Q + 1;                 // LDS ring slots (16 samples each)
  constexpr int NSLOT = 16 + Q;            // slots a pair is asked for in all (the lags need 256 + 40 samples: 18.5)
  constexpr int RS = 32 * M + 4;           // dwords per pair row
  static_assert(4 * (NL - 1) < AH && M == 4, "ring reach");
  __shared__ __align__(16) float lds_all[K4J_WAVES][PPW * RS];
  __shared__ __align__(16) float2 ptl_all[K4J_WAVES][8 * K4J_PT_STRIDE];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *lds = lds_all[wv];
  float2 *ptl = ptl_all[wv];

  const long long total = (long long)G * UWSPR_NSYM;
  const unsigned lblock = xcd_swizzle(blockIdx.x, gridDim.x);
  const long long g0 = ((long long)lblock * K4J_WAVES + wv) * PPW;
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
  const int selA = okA ? (A.nvalid >> 16) & 0xff : 0, selB = okB ? (Bg.nvalid >> 16) & 0xff : 0;
  const bool liveB = sb < PPW && okB;
  const bool use_tab = ptab != nullptr && (okA || liveB) && (!okA || selA != 0) && (!liveB || selB != 0);

  const int pr = lane >> 2;
  const int tone = lane & 3;
  const bool mineA = pr < sb;
  const int own_i = mineA ? iA0 + pr : pr - sb;
  // nothing live in this wavefront: zeros for the dead groups' hypotheses, done
  if (!okA && !liveB) {   // wave-uniform
    if (g0 + pr < total) {
      const dev_grp &gy = mineA ? A : Bg;
      for (int l = 0; l < (gy.nvalid & 0xff) && l < NL; l++)
        p_out[((long long)(gy.hyp_base + (int)((gy.hmap >> (4 * l)) & 15u)) * UWSPR_NSYM + own_i) * 4 + tone] = 0.0f;
    }
    return;
  }
  // (a live group without a table -- stage 2 gave the candidate a drift: the tone frequency depends on the symbol -- puts
  // the wavefront on per-lane phasor recurrences, cc:193-195: TAB = false below)
  const float2 *tabA = ptab, *tabB = ptab;
  if (use_tab) {
    tabA = ptab + ((size_t)(gA / gps) * kPtabPerSlot + (max(selA, 1) - 1)) * kPtabFloat2;
    tabB = ptab + ((size_t)((gA + 1) / gps) * kPtabPerSlot + (max(selB, 1) - 1)) * kPtabFloat2;
    if (!selA) tabA = tabB;     // a dead group's lanes read some valid table (their results are discarded)
    if (!selB) tabB = tabA;
  }
  // The lag slot nobody needs: slot 2 when it repeats the stage-4 winner (nvalid bit 8) in every live group of the
  // wavefront; else slot 5 when every live group holds at most five tries; else none.
  const bool knownA = !okA || (A.nvalid & 0x100) != 0, knownB = !okB || (Bg.nvalid & 0x100) != 0;
  const bool skip_mid = knownA && (knownB || sb >= PPW);
  const bool skip_last = !skip_mid && (!okA || nvA <= 5) && (!liveB || nvB <= 5);
  // everything from here on once per variant of the left-out lag slot (SKIPL: 2 known, 5 unused, -1 none), nothing
  // shared between the three: 12 KB of code each
  auto body = [&](auto skip_tag, auto tab_tag) __attribute__((always_inline)) {
  constexpr int SKIPL = decltype(skip_tag)::value;
  constexpr bool TAB = decltype(tab_tag)::value;       // phasors from the slot's table (else: this lane's recurrence)
  const int l0A = okA ? A.lag[0] : 1 - 256 * iA0;   // dead groups point at safe samples
  const int l0B = okB ? Bg.lag[0] : 1;

  const int own_nb = (mineA ? l0A : l0B) + 256 * own_i;
  const bool interior = __all((own_nb > 0) && (own_nb + 255 + 16 * Q < np)  /* the loader fetches whole slots */);

  // ---- cooperative loader: load j of a slot = pair 4j + lane/16, sample lane%16 ----
  const int kk = lane & 15;
  const int segq = lane >> 4;
  const float2 *src[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pj = 4 * j + segq;
    const bool sA = pj < sb;
    src[j] = frames + (long long)(sA ? frA : frB) * fstride + ((sA ? l0A + 256 * (iA0 + pj) : l0B + 256 * (pj - sb)) + kk);
  }
  float2 stage[4];
  auto load_slot = [&](int sl) {
    if (interior) {
#pragma unroll
      for (int j = 0; j < 4; j++) stage[j] = src[j][16 * sl];
    } else {   // a frame edge: indices from scratch (nothing of this path stays in registers)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int pj = 4 * j + segq;
        const bool sA = pj < sb;
        const int n = (sA ? l0A + 256 * (iA0 + pj) : l0B + 256 * (pj - sb)) + kk + 16 * sl;
        const bool inr = (n > 0) && (n < np);  // cc:205, sample 0 excluded
        const float2 v = frames[(long long)(sA ? frA : frB) * fstride + min(max(n, 0), np - 1)];
        stage[j] = inr ? v : make_float2(0.0f, 0.0f);
      }
    }
  };
  const int st_row = segq * RS + 2 * kk;
  auto store_slot = [&](int pos) {   // pos = ring position (slot mod M), wave-uniform
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<float2 *>(&lds[st_row + 4 * j * RS + 32 * pos]) = stage[j];
  };
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
      ptl[(e >> 4) * K4J_PT_STRIDE + (e & 15)] = tstage[j];
    }
  };

  // prologue: slots 0..Q and the first table slice, all loads in flight together
  {
    float2 pro[Q + 1][4];
#pragma unroll
    for (int sl = 0; sl <= Q; sl++) {
      load_slot(sl);
#pragma unroll
      for (int j = 0; j < 4; j++) pro[sl][j] = stage[j];
    }
    if (TAB) load_tab(0);
#pragma unroll
    for (int sl = 0; sl <= Q; sl++) {
#pragma unroll
      for (int j = 0; j < 4; j++) stage[j] = pro[sl][j];
      store_slot(sl);
    }
    if (TAB) store_tab();
  }
  // without a table: this lane's tone phasor step (binary64 angle, cc:173-189) and the recurrence state
  float rc = 1.0f, rs = 0.0f, rcd = 1.0f, rsd = 0.0f;
  if (!TAB) {
    const dev_grp &gy = mineA ? A : Bg;
    const float fp = (gy.m_type == UWSPR_LINEAR)
                         ? (float)((double)gy.f0 + ((double)gy.drift / 2.0) * ((double)(float)own_i - 81.0) / 81.0)
                         : gy.f0 + gy.slmc;
    const float delta = ((float)tone - 1.5f) * 1.46484375f;
    double sn, cs;
    sincos(kTwoPiDt * (double)(fp + delta), &sn, &cs);
    rcd = (float)cs;
    rsd = (float)sn;
  }

  // ring positions of slots c..c+Q as this lane's row addresses (dword offsets into `lds`)
  int sa[M];
#pragma unroll
  for (int q = 0; q < M; q++) sa[q] = pr * RS + 32 * q;
  const int trow = ((mineA ? 0 : 4) + tone) * K4J_PT_STRIDE;   // this lane's row of the slice

  float inp[NL], quad[NL];
#pragma unroll
  for (int l = 0; l < NL; l++) { inp[l] = 0.0f; quad[l] = 0.0f; }

  float4 ring[R];
  wave_lds_fence();
#pragma unroll
  for (int e = 0; e < AH; e++) ring[e] = *reinterpret_cast<const float4 *>(&lds[sa[(2 * e) >> 4] + 2 * ((2 * e) & 15)]);
#pragma unroll
  for (int e = AH; e < R; e++) ring[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

  int wpos = 0;  // ring position that slot c + Q + 1 will overwrite (= position of slot c)
  auto chunk = [&](auto phase_tag, int ch) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase_tag)::value;       // chunk number mod 3: where the register ring stands
    // in flight during the chunk's arithmetic (the last chunks re-fetch the last slot: no branch here)
    load_slot(min(ch + Q + 1, NSLOT - 1));
    if (TAB) load_tab(min(ch + 1, 15));
    wave_lds_fence();                      // the slots written so far are visible
    float4 phn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (TAB) phn = *reinterpret_cast<const float4 *>(&ptl[trow]);   // (c, s) of steps 2i, 2i + 1: one iteration ahead
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j0 = 8 * PH + i;                          // iteration number mod R (a constant after unrolling)
      const int o = 2 * i + 2 * AH;                       // the new pair's first sample, relative to the chunk
      ring[(j0 + AH) % R] = *reinterpret_cast<const float4 *>(&lds[sa[o >> 4] + 2 * (o & 15)]);
      float4 ph = phn;
      if (TAB) {
        if (i < 7) phn = *reinterpret_cast<const float4 *>(&ptl[trow + 2 * i + 2]);
      } else {
        ph.x = rc; ph.y = rs;
        k4_rot<false>(rc, rs, rcd, rsd);     // cc:193-195
        ph.z = rc; ph.w = rs;
        k4_rot<false>(rc, rs, rcd, rsd);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 0; l < NL; l++) {
        if (l == SKIPL) continue;
        const float4 v = ring[(j0 + 4 * l) % R];
        k4_mac<false>(inp[l], quad[l], v.x, v.y, ph.x, ph.y);   // cc:206-207, step 2 (8 ch + i)
        k4_mac<false>(inp[l], quad[l], v.z, v.w, ph.z, ph.w);   // step 2 (8 ch + i) + 1
      }
      // The iterations are kept apart: left alone the compiler sinks a chunk's arithmetic below all of its LDS reads
      // (sched_barrier holds the machine scheduler, not the IR passes in front of it) and the ring's 24 live pairs
      // become 40: 650 bytes of spills per lane.  An empty asm that "modifies" the accumulators and memory pins the
      // arithmetic in front of the next iteration's reads.
      asm volatile("" : "+v"(inp[0]), "+v"(inp[1]), "+v"(inp[2]), "+v"(inp[3]), "+v"(inp[4]), "+v"(inp[5]),
                        "+v"(quad[0]), "+v"(quad[1]), "+v"(quad[2]), "+v"(quad[3]), "+v"(quad[4]), "+v"(quad[5])
                   :: "memory");
      __builtin_amdgcn_sched_barrier(0);
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
  };
  for (int t = 0; t < 5; t++) {
    chunk(std::integral_constant<int, 0>{}, 3 * t);
    chunk(std::integral_constant<int, 1>{}, 3 * t + 1);
    chunk(std::integral_constant<int, 2>{}, 3 * t + 2);
  }
  chunk(std::integral_constant<int, 0>{}, 15);

  if (g0 + pr < total) {
    const int nv = mineA ? nvA : nvB;
    const int hb = mineA ? A.hyp_base : Bg.hyp_base;
    const uint32_t hm = mineA ? A.hmap : Bg.hmap;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      if (l < nv && l != SKIPL) {
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
  };
  if (use_tab) {                                                                       // wave-uniform, all of it
    if (skip_mid) body(std::integral_constant<int, 2>{}, std::true_type{});
    else if (skip_last) body(std::integral_constant<int, 5>{}, std::true_type{});
    else body(std::integral_constant<int, -1>{}, std::true_type{});
  } else {
    if (skip_mid) body(std::integral_constant<int, 2>{}, std::false_type{});
    else if (skip_last) body(std::integral_constant<int, 5>{}, std::false_type{});
    else body(std::integral_constant<int, -1>{}, std::false_type{});
  }
}

// groups: lags lag[0] + 8 l, l < nvalid <= 6 (the schedule's S5 emitter, eager tries)
void launch_tonecorr_jig(uwspr_ctx *c, const float *frames, int B, const dev_grp *grps, int G, int64_t nhyps, float4 *p,
                         int gps) {
  if (G <= 0) return;
  prof_scope ps(c, UWSPR_K_TONECORR, nhyps, true);
  const long long total = (long long)G * UWSPR_NSYM;
  const long long waves = (total + 15) / 16;
  const unsigned blocks = (unsigned)((waves + K4J_WAVES - 1) / K4J_WAVES);
  if (gps < 1) gps = 1;
  launch_timed(c, ps, k4_jig, dim3(blocks), dim3(64 * K4J_WAVES), 0, (const float2 *)frames, c->fstride, c->np, B, grps, G,
               (float *)p, c->use_ptab ? (const float2 *)c->d_ptab : nullptr, gps);
}

}  // namespace uwspr
