"""Single-frame API, stack-mode execution: a stream of independent frames served at the batch rates of the MI355X.

The reference evaluates one frame per `model(...)` call (`evaluation/eval_all.py:63-131`, `data/options.py:46` val_batch_size = 1).  On
MI355X one KITTI frame is a chain of ~250 dependent launches that cannot fill 256 CUs; B frames through the SAME launches (stack mode,
`CoFiI2P.stack_frames`) can: 480 frames/s with one frame per submission, 525 / 580 / 598 / 635 with 2 / 4 / 8 / 16 (bf16x6, DESIGN.md
section 12).  `FrameBatcher` keeps the caller's side at one frame per call:

    fb = FrameBatcher(model, batch=16)
    for pyr, img in frames:                 # device-resident pyramid dict (int32 or int64 tables) + (1, 3, H, W) image, equal sizes
        t = fb.submit(pyr, img)             # copies the frame into the stack being filled (one batched copy launch); returns at once
        ...
        out8 = fb.result(t)                 # the reference's 8-tuple of THAT frame (flushes a partly filled stack if it must)

Frames are copied into a ring of static stacks (`preprocess.FrameStack`); a full stack is submitted with
`forward_async(slot, stack.pyr, stack.img, inputs_stable=True)` on one of `streams` HIP streams, so the captured hipGraph of a slot reads the
stack in place and several submissions are in flight.  A partly filled stack is completed by repeating its last frame (results of the
padding are dropped): every submission has the one shape its graph was captured for.  Outputs are views of the slot's static buffers:
valid until the stack is reused, i.e. for the next `ring - 1` submissions - clone what must live longer.  Results are those of the
stack-mode forward (per-frame statistics, frame-local gathers: `tests/test_forward_gpu.py::test_stack_mode_batch_equals_single_frames`).
"""
from typing import Dict, List, Optional, Tuple

import torch

from .network import CoFiI2P
from .preprocess import FrameStack


class FrameBatcher:
    def __init__(self, model: CoFiI2P, batch: int = 16, streams: int = 4, ring: Optional[int] = None, slot_base: int = 200):
        if batch < 1:
            raise ValueError("batch must be >= 1")
        self.model, self.B = model, int(batch)
        self.S = max(1, int(streams))
        self.ring = int(ring) if ring is not None else 2 * self.S     # stacks: one being filled + the ones in flight
        if self.ring < 2:
            raise ValueError("ring must hold at least two stacks")
        self.slot_base = slot_base
        self._streams: Optional[List[torch.cuda.Stream]] = None
        self._stacks: List[Optional[FrameStack]] = [None] * self.ring
        self._handles: List[Optional[Dict]] = [None] * self.ring     # forward_async handle of the stack's last submission
        self._results: List[Optional[list]] = [None] * self.ring     # finished 8-tuples of that submission
        self._serial = [0] * self.ring                               # submissions made from this stack: tickets of older ones are stale
        self._cur, self._fill = 0, 0                                 # stack being filled, frames in it
        self._last: Optional[Tuple[Dict, torch.Tensor]] = None
        self._nsub = 0

    # ------------------------------------------------------------------ internals
    def _stream(self, j: int) -> torch.cuda.Stream:
        if self._streams is None:
            self._streams = self.model.frame_streams(self.S)
        return self._streams[j % self.S]

    def _collect(self, j: int):
        """read out the stack's pending submission.  A frame the model cannot serve (fewer than 4 coarse matches at every threshold -
        padding frames, copies of the last real one, can be such frames too) keeps its exception as ITS result: the other frames of the
        stack keep theirs, and the handle is cleared whatever happens, so one bad frame cannot wedge the ring."""
        h = self._handles[j]
        if h is None:
            return
        try:
            res = self.model.finish(h, per_frame_errors=True)
            self._results[j] = res if isinstance(res, list) else [res]
        finally:
            self._handles[j] = None

    def _launch(self, j: int):
        st = self._stacks[j]
        with torch.cuda.stream(self._stream(j)):
            self._handles[j] = self.model.forward_async(self.slot_base + j, st.pyr, st.img, inputs_stable=True)
        self._results[j] = None
        self._nsub += 1

    # ------------------------------------------------------------------ API
    def submit(self, pc_data_dict: Dict, img: torch.Tensor) -> Tuple[int, int, int]:
        """one frame -> a ticket.  The frame's tensors are read by a copy kernel enqueued on the submission's stream before this returns
        control to the caller's NEXT enqueue on that stream only - keep them unmodified until `result()` of any later ticket, or pass
        tensors that are not rewritten (a loader ring)."""
        j = self._cur
        if self._fill == 0:
            self._collect(j)                      # the stack's previous submission must have been read out before it is overwritten
            self._serial[j] += 1
        cur = torch.cuda.current_stream(img.device)
        s = self._stream(j)
        s.wait_stream(cur)                        # the frame may have been produced on the caller's stream
        with torch.cuda.stream(s):
            # conversions (int64 tables -> int32, non-contiguous inputs -> copies) run ON the submission's stream: their temporaries are
            # blocks of that stream's allocator pool, reused only behind the copy kernel that reads them.  (Made on the caller's stream
            # they could be handed to the NEXT frame's conversions while this stream, queued behind an earlier stack's forward, had not
            # copied them yet.)
            pyr = {k: ([CoFiI2P._as_idx32(t) for t in pc_data_dict[k]] if k != "points" else [p.contiguous() for p in pc_data_dict[k]])
                   for k in FrameStack.KEYS}
            feats = pc_data_dict["feats"].contiguous()
            if self._stacks[j] is None:
                self._stacks[j] = FrameStack(pyr, feats, img, self.B)
            self._stacks[j].put(self._fill, pyr, feats, img)
        self._last = (pyr, feats, img)
        ticket = (j, self._serial[j], self._fill)
        self._fill += 1
        if self._fill == self.B:
            self._launch(j)
            self._cur, self._fill = (j + 1) % self.ring, 0
        return ticket

    def flush(self):
        """submit the partly filled stack (padded with copies of its last frame)"""
        if self._fill == 0:
            return
        j = self._cur
        pyr, feats, img = self._last
        with torch.cuda.stream(self._stream(j)):
            for f in range(self._fill, self.B):
                self._stacks[j].put(f, pyr, feats, img)
        self._launch(j)
        self._cur, self._fill = (j + 1) % self.ring, 0

    def result(self, ticket: Tuple[int, int, int]):
        """-> the reference's 8-tuple of the ticket's frame (views of the slot's static outputs; see the module docstring for their lifetime)"""
        j, serial, f = ticket
        if serial != self._serial[j]:
            raise RuntimeError("FrameBatcher: the ticket's stack has been reused (results live for ring - 1 = %d later submissions)" % (self.ring - 1))
        if j == self._cur and self._fill > 0:
            self.flush()
        self._collect(j)
        if self._results[j] is None:
            raise RuntimeError("FrameBatcher: no submission is pending for this ticket")
        r = self._results[j][f]
        if isinstance(r, Exception):
            raise r                               # this frame only: the other tickets of the stack are served
        return r

    @property
    def submissions(self) -> int:
        return self._nsub
