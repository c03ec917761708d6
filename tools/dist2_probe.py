#!/usr/bin/env python3
"""Two ranks over RCCL: the C-ABI gather (uwspr_dist_gather) against torch.distributed's, byte for byte.
Needs two GPUs (rank r on device r): on a one-GPU box RCCL refuses ('Duplicate GPU detected', tried in round 3).
    python tools/dist2_probe.py"""
import os, subprocess, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in ps:
        try:
            rc = p.wait(timeout=150) or rc
        except subprocess.TimeoutExpired:
            p.kill(); rc = 99
    print("exit", rc)
    sys.exit(rc)
import numpy as np
import torch, torch.distributed as dist
import gr_uwspr_amd as G
from gr_uwspr_amd import dist as D
rank = int(os.environ["RANK"])
ndev = torch.cuda.device_count()
dev = torch.device("cuda", rank % ndev)
torch.cuda.set_device(rank % ndev)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
print(rank, "process group up", flush=True)
frames = G.synth.make_frames(8, seed=100 + rank, snr_db=-18.0)
c = G.Context(device=rank % ndev)
fr = torch.from_numpy(frames).to(dev)
cd = torch.empty(8 * c.maxfreqs * 48, dtype=torch.uint8, device=dev); nd = torch.empty(8, dtype=torch.int32, device=dev)
od = torch.empty(8 * G.native.DEMOD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
c.pipeline_batch_into(fr, cd, nd, od, max_per_frame=1)
slab = torch.zeros((8, D.SLAB_BYTES), dtype=torch.uint8, device=dev)
c.pack_slabs_into(8, D.SLAB_K, slab); c.synchronize()
ref = D.gather_slabs(slab, dst=0)
uid = torch.zeros(128, dtype=torch.uint8, device=dev)
if rank == 0:
    uid.copy_(torch.frombuffer(bytearray(G.Context.dist_unique_id()), dtype=torch.uint8))
dist.broadcast(uid, src=0)
c.dist_init(rank, 2, bytes(uid.cpu().numpy().tobytes()))
print(rank, "C-ABI communicator up", flush=True)
recv = torch.zeros((2, 8, D.SLAB_BYTES), dtype=torch.uint8, device=dev) if rank == 0 else None
c.dist_gather(slab, recv, root=0); c.synchronize()
if rank == 0:
    print("C-ABI gather == torch gather:", bool(torch.equal(recv, ref.to(dev))), flush=True)
c.dist_finalize(); c.close()
dist.barrier(); dist.destroy_process_group()
