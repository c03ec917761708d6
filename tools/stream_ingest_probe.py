import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import gr_uwspr_amd as G
dev = torch.device("cuda", 0)
B, hop = 256, 3375
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
for st in streams:
    with torch.cuda.stream(st): torch.zeros(1, device=dev)
torch.cuda.synchronize()
os.environ["UWSPR_OPTIONS"] = "sched=" + os.environ.get("FORM", "1")
frames = G.synth.make_frames(B, seed=1, snr_db=-20.0)
cc = G.Context(); cc.set_stream(streams[0].cuda_stream)
cin = G.Context(); cin.set_stream(streams[1].cuda_stream)
cands = torch.empty(B * cc.maxfreqs * 48, dtype=torch.uint8, device=dev)
npk = torch.empty(B, dtype=torch.int32, device=dev)
out = torch.empty(B * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
rng = np.random.default_rng(3)
chunk_pg = (0.5 * rng.standard_normal((B * hop, 2))).astype(np.float32)
chunk_pin = torch.from_numpy(chunk_pg).pin_memory().numpy()
cin.stream_open(hop, B)
cin.stream_push(frames[0][:45000 - hop])
fr = [torch.empty((B, 45000, 2), dtype=torch.float32, device=dev) for _ in range(2)]
def T(f, n=8):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n
def push_take(chunk):
    cin.stream_push(chunk); cin.stream_take(B, fr[0])
print("push+take pageable ms", T(lambda: push_take(chunk_pg)))
print("push+take pinned   ms", T(lambda: push_take(chunk_pin)))
print("pipeline only      ms", T(lambda: cc.pipeline_batch_into(fr[0], cands, npk, out, max_per_frame=1)))
hb = (torch.empty(B, dtype=torch.int32).pin_memory(), torch.empty(cands.numel(), dtype=torch.uint8).pin_memory(), torch.empty(out.numel(), dtype=torch.uint8).pin_memory())
def d2h():
    with torch.cuda.stream(streams[0]):
        hb[0].copy_(npk, non_blocking=True); hb[1].copy_(cands, non_blocking=True); hb[2].copy_(out, non_blocking=True)
print("d2h async pinned   ms", T(d2h))
print("d2h .cpu()         ms", T(lambda: (npk.cpu(), cands.cpu(), out.cpu())))
ev = [torch.cuda.Event() for _ in range(4)]
def both(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        cin.stream_push(chunk_pin); cin.stream_take(B, fr[k]); ev[k].record(streams[1])
    with torch.cuda.stream(streams[0]):
        streams[0].wait_event(ev[k])
        cc.pipeline_batch_into(fr[k], cands, npk, out, max_per_frame=1)
print("push/take || pipeline ms", T(both, 16))
def both_d2h(i=[0]):
    both(); d2h()
print("  + d2h               ms", T(both_d2h, 16))
def both_noev(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        cin.stream_push(chunk_pin); cin.stream_take(B, fr[k])
    with torch.cuda.stream(streams[0]):
        cc.pipeline_batch_into(fr[k ^ 1], cands, npk, out, max_per_frame=1)
print("no events             ms", T(both_noev, 16))
evt = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
def both_timing_events(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        cin.stream_push(chunk_pin); cin.stream_take(B, fr[k]); evt[k].record(streams[1])
    with torch.cuda.stream(streams[0]):
        streams[0].wait_event(evt[k])
        cc.pipeline_batch_into(fr[k], cands, npk, out, max_per_frame=1)
print("timing-enabled events ms", T(both_timing_events, 16))
def ev_only(i=[0]):
    k = i[0] & 1; i[0] += 1
    ev[k].record(streams[1]); streams[0].wait_event(ev[k])
print("record+wait alone     ms", T(ev_only, 16))
def ev_with_tiny(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        a = torch.zeros(8, device=dev); ev[k].record(streams[1])
    with torch.cuda.stream(streams[0]):
        streams[0].wait_event(ev[k]); b = torch.zeros(8, device=dev)
print("tiny kernels + events ms", T(ev_with_tiny, 16))
def take_then_pipeline_events_only_take(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        cin.stream_push(chunk_pin); cin.stream_take(B, fr[k]); ev[k].record(streams[1])
    with torch.cuda.stream(streams[0]):
        streams[0].wait_event(ev[k]); b = torch.zeros(8, device=dev)
print("ingest + event + tiny ms", T(take_then_pipeline_events_only_take, 16))
def tiny_then_pipeline(i=[0]):
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(streams[1]):
        a = torch.zeros(8, device=dev); ev[k].record(streams[1])
    with torch.cuda.stream(streams[0]):
        streams[0].wait_event(ev[k])
        cc.pipeline_batch_into(fr[k], cands, npk, out, max_per_frame=1)
print("tiny + event + pipeline ms", T(tiny_then_pipeline, 16))
