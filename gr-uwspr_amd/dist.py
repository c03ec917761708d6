"""Multi-GPU: frames are independent units (FDR_impl.cc:214 and
sync_and_demodulate_impl.cc:315 read only their own PDU), so the path shards
data-parallel over frames with NO data-path collective.  The one exchange is the
final gather of fixed-size per-frame slabs {npk, candidate_t[K], refined
(f1, shift1, drift1, sync1)} to rank 0 -- torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Round-robin sharding (BASELINE configs[3]): global frame b lives on rank
b mod G at local index b div G; rank 0 restores global order after the gather.
"""
import numpy as np

SLAB_K = 8                      # candidates kept per frame in the gathered slab
SLAB_BYTES = 16 + SLAB_K * 48 + 16   # npk + pad | candidate_t[K] | f1, shift1, drift1, sync1


def shard_indices(total, rank, world):
    """Global frame indices owned by `rank` (round-robin)."""
    return np.arange(rank, total, world)


def local_count(total, rank, world):
    return (total - rank + world - 1) // world


def pack_slabs(cands_u8, npk_i32, demod_u8, maxfreqs, per_frame, demod_itemsize):
    """Device-side (torch) packing of the per-frame slab from the library's output
    buffers: cands_u8 [B*maxfreqs*48] uint8, npk [B] int32, demod [B*per_frame*itemsize]."""
    import torch
    B = npk_i32.numel()
    slab = torch.zeros((B, SLAB_BYTES), dtype=torch.uint8, device=npk_i32.device)
    slab[:, 0:4] = npk_i32.view(torch.uint8).view(B, 4)
    k = min(SLAB_K, maxfreqs)
    slab[:, 16:16 + k * 48] = cands_u8.view(B, maxfreqs * 48)[:, :k * 48]
    slab[:, 16 + SLAB_K * 48:] = demod_u8.view(B, per_frame * demod_itemsize)[:, :16]
    return slab


def gather_slabs(slab, dst=0):
    """Equal-sized shards -> rank `dst` gets [world, B, SLAB_BYTES]; others get None.
    With world_size 1 this is the identity (no collective)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return slab.unsqueeze(0)
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and slab.is_cuda:
        slab = slab.cpu()          # CPU rehearsal of the N>1 path; RCCL gathers device memory
    if dist.get_rank() == dst:
        out = [torch.empty_like(slab) for _ in range(world)]
        dist.gather(slab, gather_list=out, dst=dst)
        return torch.stack(out)
    dist.gather(slab, gather_list=None, dst=dst)
    return None


def restore_order(gathered, total):
    """[world, Bl, ...] (round-robin shards, padded to equal Bl) -> [total, ...] in
    global frame order b = local_idx*G + rank."""
    world, bl = gathered.shape[0], gathered.shape[1]
    out = gathered.transpose(0, 1).reshape(world * bl, *gathered.shape[2:])
    return out[:total]


def unpack_slab(slab_np, cand_dtype):
    """numpy view of one gathered slab row -> (npk, candidates[K], f1, shift1, drift1, sync1)."""
    npk = int(np.frombuffer(slab_np[:4].tobytes(), np.int32)[0])
    cands = np.frombuffer(slab_np[16:16 + SLAB_K * 48].tobytes(), cand_dtype)
    tail = slab_np[16 + SLAB_K * 48:].tobytes()
    f1, drift1, sync1 = np.frombuffer(tail[:12], np.float32)
    shift1 = int(np.frombuffer(tail[12:16], np.int32)[0])
    return npk, cands[:min(npk, SLAB_K)], float(f1), shift1, float(drift1), float(sync1)
