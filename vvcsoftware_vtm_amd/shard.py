"""Multi-GPU partitioning of the hot path: random-access intra periods are self-contained, so rank r
processes the pictures of intra periods r, r+world, ...; the only data-path exchange is ONE reconstructed boundary
(CRA) picture per chunk hand-over, point-to-point (RCCL send/recv over one xGMI link; gloo in the CPU tests).
No all-reduce / ring collective exists anywhere on the path."""
import hashlib

import torch
import torch.distributed as dist

INTRA_PERIOD = 32


def chunk_assignment(n_pictures, world, intra_period=INTRA_PERIOD):
    """-> list (per rank) of lists of (first_poc, last_poc_exclusive) chunks, in coding order."""
    chunks = [(s, min(s + intra_period, n_pictures)) for s in range(0, n_pictures, intra_period)]
    return [chunks[r::world] for r in range(world)]


def boundary_owner(chunk_index, world):
    return chunk_index % world


def exchange_boundary(planes, rank, world, tag=0):
    """Ring hand-over of the reconstructed boundary picture: rank r sends its planes to (r+1) % world and receives the
    planes of (r-1) % world.  Returns the received planes (same shapes).  world == 1: returns the input unchanged."""
    if world == 1:
        return planes
    recv = [torch.empty_like(p) for p in planes]
    ops = []
    for p, q in zip(planes, recv):
        ops.append(dist.P2POp(dist.isend, p.contiguous(), (rank + 1) % world))
        ops.append(dist.P2POp(dist.irecv, q, (rank - 1) % world))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return recv


def picture_hash(planes):
    h = hashlib.md5()
    for p in planes:
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def gather_hashes(local_hashes, world):
    """every rank contributes {poc: md5}; rank 0 gets the merged dict (object gather, control path only)."""
    if world == 1:
        return dict(local_hashes)
    out = [None] * world
    dist.all_gather_object(out, local_hashes)
    merged = {}
    for d in out:
        merged.update(d)
    return merged


def install_reference(planes, ref_padded, margins):
    """Chunk hand-over, receiving side: the boundary picture becomes a reference picture of this rank's next chunk -- copied into
    the padded DPB slot `ref_padded` (one 2-D int16 tensor per component, margins (mx, my) per component) and border-extended on
    the device (Picture::extendPicBorder, vvcgpu_extend_border).  Device tensors only: there is no host form of this step."""
    from . import ops
    for p, r, (mx, my) in zip(planes, ref_padded, margins):
        h, w = p.shape
        r[my:my + h, mx:mx + w].copy_(p)
        ops.extend_border(r, mx, my)
