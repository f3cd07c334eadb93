"""Multi-GPU partitioning of the hot path: random-access intra periods are self-contained, so rank r
processes the pictures of intra periods r, r+world, ...; the only data-path exchange is ONE reconstructed boundary
(CRA) picture per chunk hand-over, point-to-point (one grouped RCCL send/recv per hand-over over one xGMI link; gloo in the CPU tests).
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


def _backend(group=None):
    return str(dist.get_backend(group)).lower() if dist.is_available() and dist.is_initialized() else None


def _unbatched(op, tensor, peer, group=None):
    """dist.isend / dist.irecv as single operations -- gloo only.  On an RCCL group (initialised with `device_id=`, i.e. eagerly) an unbatched
    point-to-point operation is serialised with every other operation of the communicator: an early receive would sit in front of the rank's own
    send on every rank of the ring and nobody would ever send.  Refused here so that it cannot come back by accident."""
    if _backend(group) != "gloo":
        raise RuntimeError("shard: unbatched %s on backend %r -- point-to-point on RCCL goes through ONE dist.batch_isend_irecv per hand-over"
                           % (op.__name__, _backend(group)))
    return op(tensor, peer, group=group)


def _wire(t):
    """the tensor as the backend sees it: RCCL has no 16-bit integer type (ProcessGroupNCCL refuses `Short`), so sample planes travel as the bytes
    they are -- a uint8 view of the same memory, no copy"""
    return t if t.dtype == torch.uint8 else t.view(torch.uint8)


class Handover:
    """One chunk hand-over in flight (the exchange unit of the reference's file-based analogue, EncGOP.cpp:1146-1300 DebugBitstream re-entry; the state
    that travels beside the picture is EncCu.h:119-122, `SIDE_RECORD`).

    `post_recv` at the start of a chunk prepares the receive buffers, `send` is called when the boundary picture exists, `wait` only in front of the
    first use of the received picture (the first motion compensation of the next chunk: `Workload.run_gpu(pre_mc=...)`) -- a rank never waits at a
    step boundary, only for data it is about to read.  How the operations reach the backend:

    * RCCL (`nccl`): nothing is posted early.  `send` issues the receives from rank r - 1 and the sends to rank r + 1 of this hand-over as ONE
      grouped `dist.batch_isend_irecv` (ncclGroupStart .. End): inside a group the communicator's stream order does not put a rank's receive in
      front of its own send, so the ring cannot wait on itself at any world size (2 included, where both directions share one peer).  The transfer
      (24.9 MB at 4K, ~0.16 ms on one xGMI link) runs on the communicator's stream behind the kernels that produced the picture and beside the next
      chunk's searches.
    * gloo (CPU tests): no stream order exists, the receives are really posted at chunk start and the sends are single operations.
      `batched=True` forces the grouped form on gloo too, so the CPU tests walk the code path RCCL takes.

    world == 1: the picture is handed to the rank itself without communication -- unless `loopback` is set, which sends it to rank 0 itself through
    the backend (RCCL accepts a grouped send + receive to the own rank): the one-GPU test of the tensor / stream handling of the N > 1 path.
    The side record travels with the planes."""

    def __init__(self, like_planes, rank, world, with_record=True, group=None, batched=None, loopback=False):
        self.rank, self.world = rank, world
        self.recv_planes = None
        self.recv_record = None
        self.reqs = []
        self.like = like_planes
        self.with_record = with_record
        self.group = group
        self.comm = world > 1 or loopback
        self.batched = (_backend(group) != "gloo") if batched is None else bool(batched)
        if self.comm and not self.batched and _backend(group) != "gloo":
            raise RuntimeError("shard.Handover: batched=False needs the gloo backend (have %r)" % _backend(group))
        self.issued = []                                  # what went to the backend, in order: ("batch", n_ops) | ("irecv", 1) | ("isend", 1)

    def post_recv(self):
        if not self.comm:
            return self
        src = (self.rank - 1) % self.world
        self.recv_planes = [torch.empty_like(p) for p in self.like]
        if self.with_record:
            self.recv_record = torch.empty(SIDE_RECORD.itemsize, dtype=torch.uint8, device=self.like[0].device)
        if not self.batched:
            for q in self.recv_planes + ([self.recv_record] if self.with_record else []):
                self.reqs.append(_unbatched(dist.irecv, _wire(q), src, self.group))
                self.issued.append(("irecv", 1))
        return self

    def send(self, planes, record=None):
        if not self.comm:
            self.recv_planes = planes
            self.recv_record = None if record is None else side_record_tensor(record, planes[0].device)
            return self
        if self.recv_planes is None:
            self.post_recv()
        src, dst = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        self._keep = [p.contiguous() for p in planes]                 # alive (and unmodified by the caller) until wait()
        if self.with_record:
            self._keep.append(side_record_tensor(record if record is not None else empty_side_record(), planes[0].device))
        if self.batched:
            recvs = self.recv_planes + ([self.recv_record] if self.with_record else [])
            ops = [dist.P2POp(dist.irecv, _wire(q), src, self.group) for q in recvs] + [dist.P2POp(dist.isend, _wire(p), dst, self.group) for p in self._keep]
            self.reqs += dist.batch_isend_irecv(ops)
            self.issued.append(("batch", len(ops)))
        else:
            for p in self._keep:
                self.reqs.append(_unbatched(dist.isend, _wire(p), dst, self.group))
                self.issued.append(("isend", 1))
        return self

    def wait(self):
        """makes the CURRENT stream wait for the transfer (RCCL) / blocks until it is done (gloo); returns (planes, record)"""
        for r in self.reqs:
            r.wait()
        self.reqs = []
        self._keep = None
        rec = None if self.recv_record is None else side_record_from_tensor(self.recv_record)
        return self.recv_planes, rec


def exchange_boundary(planes, rank, world, tag=0, record=None, return_record=False, batched=None, loopback=False):
    """Ring hand-over of the reconstructed boundary picture (+ the side record): rank r sends to (r+1) % world and receives from (r-1) % world.
    Synchronous form (post, send, wait in one call); the asynchronous form is `Handover`.  Returns the received planes (and the record when asked)."""
    h = Handover(planes, rank, world, with_record=True, batched=batched, loopback=loopback).post_recv().send(planes, record)
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
