#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ (run in the build container,
where /root/reference exists).  Everything written here is DATA: inputs and
expected outputs.  Sources of truth:

  slm_qa.txt, slm_table_ref.npy, fano_ref.npz, mettab_ref.npy
      outputs of the REAL reference objects built by oracle/Makefile `ref`
      (lib/slm.cc, lib/slm_qa.cc, lib/Fano.cc, lib/helpers.cc), run here.
  ve3emb_known.json
      the known answers the survey recorded from the real reference
       for examples/VE3EMB.c2 -- hand-entered, not
      computed by this script; the script only re-checks the oracle against them.
  oracle_vectors.npz
      outputs of oracle/uwspr_oracle.c (our restatement) on seeded synthetic
      frames; they let the GPU tests run against committed vectors as well as
      against the live oracle.
  VE3EMB.c2
      copy of the reference's example data file examples/VE3EMB.c2.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py as O  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402


def main():
    O.build(ref=True)
    # ---- real reference: SLM ------------------------------------------------
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "slm_qa")], capture_output=True,
                         text=True, check=True).stdout
    open(os.path.join(HERE, "slm_qa.txt"), "w").write(out)
    R = O.ref()
    import ctypes as C
    V1 = (C.c_double * 200)(); V2 = (C.c_double * 200)()
    p1 = (C.c_int * 200)(); p2 = (C.c_int * 200)()
    n = R.ref_slm_generate_all(V1, V2, p1, p2, 200)
    inst = np.array([(V1[i], V2[i], p1[i], p2[i]) for i in range(n)], np.float64)
    tab = np.zeros((n, 2, 120), np.float32)
    for i in range(n):
        for ci, cf in enumerate((1500.0, 3000.0)):
            for t in range(120):
                tab[i, ci, t] = R.ref_slm_frequency_drift(V1[i], V2[i], p1[i], p2[i], cf, float(t))
    np.save(os.path.join(HERE, "slm_instances_ref.npy"), inst)
    np.save(os.path.join(HERE, "slm_table_ref.npy"), tab)
    np.save(os.path.join(HERE, "mettab_ref.npy"), O.ref_mettab())

    # ---- real reference: Fano on soft symbols of noisy synthetic frames -------
    frames, meta = G.synth.make_frames(6, seed=1234, snr_db=-24.0, return_meta=True)
    fdr = O.FDR()
    syms, rcs, datas, cycles, metrics = [], [], [], [], []
    for b in range(frames.shape[0]):
        cands = fdr.transform(frames[b])
        if len(cands) == 0:
            continue
        d = O.demod_candidate(cands[0], 1500, frames[b])
        if not d["worth_a_try"]:
            continue
        for idt in (0, 1, 2, 7, 16):
            s = O.deinterleave(d["symbols"][idt])
            rc, data, metric, cyc = O.ref_fano_decode(s)
            syms.append(d["symbols"][idt]); rcs.append(rc); datas.append(data)
            cycles.append(cyc); metrics.append(metric)
    # plus pure-noise vectors that must time out
    rng = np.random.default_rng(7)
    for _ in range(3):
        s = rng.integers(0, 256, size=162).astype(np.uint8)
        rc, data, metric, cyc = O.ref_fano_decode(O.deinterleave(s))
        syms.append(s); rcs.append(rc); datas.append(data); cycles.append(cyc); metrics.append(metric)
    enc_in = rng.integers(0, 256, size=(4, 11)).astype(np.uint8)
    enc_out = np.stack([O.ref_fano_encode(e) for e in enc_in])
    msgs = np.array([[0xd4 - 256, 0x2c, 0x73, 0xeb - 256, 0x3a, 0x77, 0x80 - 256]], np.int8)
    texts = [O.ref_unpk(m) for m in msgs]
    for dat in datas:
        if True:
            m = np.array([int(x) - 256 if x > 127 else int(x) for x in dat[:7]], np.int8)
            msgs = np.vstack([msgs, m[None]])
            texts.append(O.ref_unpk(m))
    np.savez_compressed(os.path.join(HERE, "fano_ref.npz"), symbols=np.array(syms), rc=np.array(rcs),
                        data=np.array(datas), cycles=np.array(cycles), metric=np.array(metrics),
                        enc_in=enc_in, enc_out=enc_out, msgs=msgs, texts=np.array(texts))

    # ---- oracle vectors on seeded synthetic frames -----------------------------
    frames = G.synth.make_frames(4, seed=0xC0FFEE, snr_db=-20.0)
    vec = {}
    cand_all, npk, grids, demods = [], [], [], []
    for b in range(4):
        ps = fdr.spectrogram(frames[b])
        psavg, smraw, smspec, noise = fdr.stats(ps)
        pk = fdr.peaks(smspec)
        res = []
        for j in range(len(pk)):
            c, grid = fdr.search(ps, pk[j], want_grid=(j == 0))
            res.append(c)
            if j == 0:
                grids.append(grid)
        cands = np.array(res, dtype=O.CAND_DTYPE)
        cand_all.append(np.pad(cands, (0, 16 - len(cands))))
        npk.append(len(cands))
        d = O.demod_candidate(cands[0], 1500, frames[b])
        demods.append(d)
        vec["smspec%d" % b] = smspec
        vec["noise%d" % b] = np.float32(noise)
        vec["psavg%d" % b] = psavg
    vec["cands"] = np.stack(cand_all)
    vec["npk"] = np.array(npk)
    vec["grid0"] = np.stack(grids)
    for k in ("f1", "drift1", "sync1", "shift1", "worth_a_try", "jig_sync", "jig_rms", "jig_shift",
              "symbols"):
        vec["demod_" + k] = np.array([d[k] for d in demods])
    # 2 frames x 200 hypotheses (config-3 grid) of the fine sweep
    hy = G.sweep_grid(vec["cands"][:2, 0], frames_idx=[0, 1]) if hasattr(G, "sweep_grid") else None
    if hy is not None:
        sync = np.zeros(hy.size, np.float32)
        sym = np.zeros((hy.size, 162), np.uint8)
        for q, h in enumerate(hy):
            cand = np.zeros(1, O.CAND_DTYPE)[0]
            cand["m_type"] = h["m_type"]
            s, _, _, sy = O.sync_and_demodulate(cand, 1500, frames[h["frame"]], float(h["f0"]), 0, 0,
                                                0.0, int(h["lag"]), 0, 0, 1, float(h["drift"]), 50, 2)
            sync[q] = s; sym[q] = sy
        vec["sweep_hyps"] = hy
        vec["sweep_sync"] = sync
        vec["sweep_symbols"] = sym
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **vec)

    # ---- re-check the oracle against the survey's known answers ----------------
    known = json.load(open(os.path.join(HERE, "ve3emb_known.json")))
    iq = O.read_c2(os.path.join(HERE, "VE3EMB.c2"))
    c = fdr.transform(iq)
    assert len(c) == known["npk"] and int(c[0]["p2"]) == known["p2"]
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
