"""Why did the same (form, streams) configuration measure 12 % apart inside one bench.py run?
Re-creates the three-lane staged configuration several times in one process, with torch pool
streams and with the contexts' own streams, and prints ms/step of each incarnation."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gr_uwspr_amd as G  # noqa: E402

os.environ["UWSPR_OPTIONS"] = "sched=0"
dev = torch.device("cuda", 0)
B, K, NS = 256, 100, 3
batches = [G.synth.make_frames_torch(B, dev, seed=0xC0FFEE + 104729 * k, snr_db=-20.0) for k in range(5)]
slab = torch.zeros((K, B, 416), dtype=torch.uint8, device=dev)


def lanes(own):
    out = []
    for _ in range(NS):
        cx = G.Context(device=0)
        st = None
        if not own:
            st = torch.cuda.Stream(device=dev)
            cx.set_stream(st.cuda_stream)
        out.append({"ctx": cx, "st": st,
                    "cands": torch.empty(B * cx.maxfreqs * 48, dtype=torch.uint8, device=dev),
                    "npk": torch.empty(B, dtype=torch.int32, device=dev),
                    "out": torch.empty(B * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)})
    return out


def run(ls):
    def step(i):
        ln = ls[i % NS]
        ln["ctx"].pipeline_batch_into(batches[i % 5], ln["cands"], ln["npk"], ln["out"], max_per_frame=1)
        ln["ctx"].pack_slabs_into(B, 8, slab[i % K])
    for i in range(10):
        step(i)
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        for ln in ls:
            ln["ctx"].synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        for ln in ls:
            ln["ctx"].synchronize()
        torch.cuda.synchronize()
        res.append(1e3 * (time.perf_counter() - t0) / K)
    return res


def show(tag, ls, r):
    print("%-46s ms/step %s" % (tag, " ".join("%.4f" % x for x in r)), flush=True)


ls = lanes(False)
show("first contexts, first torch streams", ls, run(ls))
keep_streams = [ln["st"] for ln in ls]
for ln in ls:                                   # same contexts (same memory), NEW streams
    ln["st"] = torch.cuda.Stream(device=dev)
    ln["ctx"].set_stream(ln["st"].cuda_stream)
show("first contexts, second torch streams", ls, run(ls))
for ln in ls:
    ln["ctx"].set_stream(None)
show("first contexts, their own streams", ls, run(ls))
for ln, st in zip(ls, keep_streams):
    ln["st"] = st
    ln["ctx"].set_stream(st.cuda_stream)
show("first contexts, first torch streams again", ls, run(ls))
ls2 = lanes(False)                              # second set of contexts while the first is alive
for ln, st in zip(ls2, keep_streams):
    ln["st"] = st
    ln["ctx"].set_stream(st.cuda_stream)
show("second contexts (first alive), first streams", ls2, run(ls2))
show("first contexts once more", ls, run(ls))
for ln in ls + ls2:
    ln["ctx"].close()
ls3 = lanes(False)
for ln, st in zip(ls3, keep_streams):
    ln["st"] = st
    ln["ctx"].set_stream(st.cuda_stream)
show("third contexts (others closed), first streams", ls3, run(ls3))
