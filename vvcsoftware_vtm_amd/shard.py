"""Multi-GPU partitioning of the hot path: random-access intra periods are self-contained, so rank r
processes the pictures of intra periods r, r+world, ...; the only data-path exchange is ONE reconstructed boundary
(CRA) picture per chunk hand-over, point-to-point (RCCL send/recv over one xGMI link; gloo in the CPU tests).
No all-reduce / ring collective exists anywhere on the path."""
import hashlib

import numpy as np
import torch
import torch.distributed as dist

INTRA_PERIOD = 32

# Chunk hand-over SIDE RECORD: the encoder state outside the decoded picture buffer that makes the worker of the next chunk reproduce the sequential
# encoder byte for byte -- the per-temporal-layer ATMVP statistics of EncCu (EncoderLib/EncCu.h:119-122: m_subMergeBlkSize[10], m_subMergeBlkNum[10],
# m_prevPOC, m_clearSubMergeStatic; filled by CABACWriter.cpp:543-552, read by EncSlice.cpp:1250-1306 to choose the slice's ATMVP sub-block size).
# 88 bytes; it travels with the boundary picture (profiles/r02_chunk_exactness.txt: without it the stitched stream is valid but differs from the
# sequential one as soon as SubPuMvp is on; tests/test_chunk_stitch.py: with it the re-entered encode equals the sequential encode).
SIDE_RECORD = np.dtype([("sub_merge_blk_size", "<u4", (10,)), ("sub_merge_blk_num", "<u4", (10,)), ("prev_poc", "<u4"), ("clear_sub_merge_static", "<u4")])
assert SIDE_RECORD.itemsize == 88


def empty_side_record():
    """state of a fresh encoder (EncCu.cpp:271-274)"""
    r = np.zeros(1, SIDE_RECORD)
    r["prev_poc"] = 0xFFFFFFFF
    return r


def side_record_tensor(rec, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1).copy()).to(device)


def side_record_from_tensor(t):
    return t.detach().cpu().numpy().view(SIDE_RECORD)


def chunk_assignment(n_pictures, world, intra_period=INTRA_PERIOD):
    """-> list (per rank) of lists of (first_poc, last_poc_exclusive) chunks, in coding order."""
    chunks = [(s, min(s + intra_period, n_pictures)) for s in range(0, n_pictures, intra_period)]
    return [chunks[r::world] for r in range(world)]


def boundary_owner(chunk_index, world):
    return chunk_index % world


class Handover:
    """One chunk hand-over in flight: the receives from rank r - 1 are posted when the chunk STARTS (`post_recv`), the sends to rank r + 1 when the
    boundary picture exists (`send`), and `wait` is called only in front of the first use of the received picture (the first motion compensation of
    the next chunk) -- a rank never waits for its neighbour at a step boundary, only for data it is about to read.  The side record travels with
    the planes.  world == 1: the picture is handed to the rank itself without any communication."""

    def __init__(self, like_planes, rank, world, with_record=True):
        self.rank, self.world = rank, world
        self.recv_planes = None
        self.recv_record = None
        self.reqs = []
        self.like = like_planes
        self.with_record = with_record

    def post_recv(self):
        if self.world == 1:
            return self
        src = (self.rank - 1) % self.world
        self.recv_planes = [torch.empty_like(p) for p in self.like]
        for q in self.recv_planes:
            self.reqs.append(dist.irecv(q, src))
        if self.with_record:
            self.recv_record = torch.empty(SIDE_RECORD.itemsize, dtype=torch.uint8, device=self.like[0].device)
            self.reqs.append(dist.irecv(self.recv_record, src))
        return self

    def send(self, planes, record=None):
        if self.world == 1:
            self.recv_planes = planes
            self.recv_record = None if record is None else side_record_tensor(record, planes[0].device)
            return self
        dst = (self.rank + 1) % self.world
        self._keep = [p.contiguous() for p in planes]                 # alive until the sends complete
        for p in self._keep:
            self.reqs.append(dist.isend(p, dst))
        if self.with_record:
            self._rec = side_record_tensor(record if record is not None else empty_side_record(), planes[0].device)
            self.reqs.append(dist.isend(self._rec, dst))
        return self

    def wait(self):
        for r in self.reqs:
            r.wait()
        self.reqs = []
        rec = None if self.recv_record is None else side_record_from_tensor(self.recv_record)
        return self.recv_planes, rec


def exchange_boundary(planes, rank, world, tag=0, record=None, return_record=False):
    """Ring hand-over of the reconstructed boundary picture (+ the side record): rank r sends to (r+1) % world and receives from (r-1) % world.
    Synchronous form (post, send, wait in one call); the asynchronous form is `Handover`.  Returns the received planes (and the record when asked)."""
    h = Handover(planes, rank, world, with_record=True).post_recv().send(planes, record)
    got, rec = h.wait()
    return (got, rec) if return_record else got


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
