"""Frame exchange of the tile-partitioned path: render -> gather -> un-swizzle, one process per GPU.

Two things shape it, both consequences of xGMI being point-to-point and of how small one rank's share of a frame is:

* **Root-weighted partition.**  Rank 0 assembles the frame, so every other rank's tiles cross exactly one link into rank 0
  while rank 0's own tiles cost no transfer.  Rank 0 therefore carries `root_weight` shares and every other rank one
  (`tiles.weighted_partition`): the weight balances render time against per-link transfer time.  `root_weight = 0` means
  "do not shard" (rank 0 renders everything) -- the right answer when a frame is cheaper to render than to ship.
  `autotune` measures a few candidates on the live job and keeps the fastest, so adding GPUs never makes a frame slower.
* **One collective per `batch` frames** (fewer, larger collectives): every rank renders `batch` consecutive frames into
  one packed buffer, then ONE `gather` moves them; the host cost of a collective call (tens of microseconds, comparable
  to a rank's share of a 1080p frame) is paid once per batch.  Two batch buffers alternate so the next batch renders while
  the previous one is on the links.

The control flow is backend-agnostic: `ops` supplies the buffers and the three device operations, so the very same code
runs over RCCL with the HIP renderer (bench.py) and over gloo with the CPU oracle standing in for the renderer
(tests/test_tiles_gloo.py).

* **12 bytes per pixel on the links.**  Every pixel of a frame has alpha 1, so a non-root rank packs its batch to 3 floats
  per pixel before the gather (one small kernel per batch) and rank 0 un-swizzles with alpha = 1: 25 % fewer bytes into the
  one GPU whose inbound links bound the whole exchange.  Bit-exact (`rgb=False` sends the float4 records as they are).

ops protocol (optional: ops.gather_parts(send, recv, nbytes) replaces torch.distributed.gather -- see HipOps):
    ops.alloc(n_px, channels=4) -> torch tensor (n_px, channels) float32 on the exchange device
    ops.pack_rgb(src, n_px, dst)                               float4 records -> 3 floats per pixel (pt_pack_rgb)
    ops.unpack_rgb(...)                                        ops.unpack for 3-float parts (pt_unpack_tiles_rgb)
    ops.set_range(first, run, stride)                          partition of this rank (pt_set_partition_ex)
    ops.render(frame_index, out)                               this rank's range of that frame -> packed tensor `out`
    ops.render_full(frame_index, frame)                        (optional) the whole frame, row-major, no tiles: used for root_weight 0
    ops.unpack(packed, offset_px, part_stride_px, n_parts, first0, run, stride, frame)   (pt_unpack_tiles_ex; packed = flat tensor)
"""
import torch
import torch.distributed as dist

from . import tiles


class TileExchange:
    def __init__(self, ops, w, h, rank, world, batch, ts=32, rehearse=False, rgb=True, frames_in_flight=1):
        """frames_in_flight: that of the renderer behind `ops`.  The two batch buffers alternate and a frame only waits for the consumer
        of its buffer from frames_in_flight - 1 calls back (pt_api.hip render_common), so a shorter batch would let a frame overwrite
        tiles that the previous batch's gather / un-swizzle has not read yet: refused."""
        if frames_in_flight > 1 and max(1, batch) < frames_in_flight - 1:
            raise ValueError(f"TileExchange: batch {batch} is shorter than frames in flight - 1 ({frames_in_flight - 1})")
        self.ops, self.w, self.h, self.rank, self.world, self.batch, self.ts = ops, w, h, rank, world, max(1, batch), ts
        self.rgb = rgb  # exchange buffers carry 3 floats per pixel
        self.rehearse = rehearse  # world == 1: still issue the collective (with nothing to receive) to rehearse the call path
        self.tile_px = ts * ts
        # buffers sized for the largest share any configuration can give this rank
        tx, ty = tiles.tile_grid(w, h, ts)
        total = tx * ty
        self.cap_root = total
        self.cap_other = tiles.range_tiles_count(w, h, 1, 1, world, ts) if world > 1 else 0  # weight 1: the even interleave
        cap_own = self.cap_root if rank == 0 else self.cap_other
        self._own_store = [ops.alloc(self.batch * cap_own * self.tile_px) for _ in range(2)]
        ch = 3 if rgb else 4
        if rank == 0:
            self.frames = [ops.alloc(w * h) for _ in range(self.batch)]
            self._gathered_store = ops.alloc(world * self.batch * self.cap_other * self.tile_px, ch) if world > 1 else None
            self._dummy_store = ops.alloc(self.batch * self.cap_other * self.tile_px, ch) if world > 1 else None
        elif rgb and world > 1:
            self._send_store = ops.alloc(self.batch * self.cap_other * self.tile_px, 3)
        self.submitted = 0
        self.configure(1)

    # ---- partition ----------------------------------------------------------------------------------------------------
    def configure(self, root_weight):
        """select the partition (all ranks must pass the same weight); buffers are re-viewed, nothing is reallocated"""
        assert self.submitted % self.batch == 0, "configure between batches only"
        self.root_weight = root_weight
        self.range = tiles.weighted_partition(self.rank, self.world, root_weight)
        self.stride = self.range[2]
        self.n_root = tiles.range_tiles_count(self.w, self.h, *tiles.weighted_partition(0, self.world, root_weight), self.ts)
        self.sharded = self.world > 1 and root_weight != 0
        # weight 0 with several ranks: rank 0 renders whole frames straight into self.frames (ops.render_full), the others idle
        self.direct = self.world > 1 and root_weight == 0 and getattr(self.ops, "render_full", None) is not None
        # every non-root rank sends the same number of tiles (the first of them owns the most; later ones are zero padded)
        self.n_other = tiles.range_tiles_count(self.w, self.h, *tiles.weighted_partition(1, self.world, root_weight), self.ts) if self.sharded else 0
        n_own = self.n_root if self.rank == 0 else self.n_other
        self.own_px = n_own * self.tile_px
        self.own = [s[: self.batch * self.own_px].view(self.batch, self.own_px, 4) for s in self._own_store]
        self.other_px = self.n_other * self.tile_px
        if self.rank == 0 and self.sharded:
            self.gathered = self._gathered_store[: self.world * self.batch * self.other_px].view(self.world, self.batch * self.other_px, 3 if self.rgb else 4)
            self.gather_list = list(self.gathered.unbind(0))
            self.dummy = self._dummy_store[: self.batch * self.other_px]
        self.ops.set_range(*self.range)

    # ---- per frame ----------------------------------------------------------------------------------------------------
    def submit(self, frame_index):
        """queue one frame; the collective + un-swizzle are issued when its batch is complete"""
        b, slot = (self.submitted // self.batch) % 2, self.submitted % self.batch
        if self.direct:
            # "do not shard": rank 0 renders the frame where it belongs -- no tiles, no un-swizzle; the job is as fast as one GPU
            if self.rank == 0:
                self.ops.render_full(frame_index, self.frames[slot])
        elif self.own_px:
            self.ops.render(frame_index, self.own[b][slot])
        self.submitted += 1
        if slot == self.batch - 1:
            self._flush(b, self.batch)

    def finish(self):
        """flush a partially filled batch (end of the run)"""
        pending = self.submitted % self.batch
        if pending:
            self._flush((self.submitted // self.batch) % 2, pending)
            self.submitted += self.batch - pending  # the next submit starts a fresh batch

    def _flush(self, b, n_frames):
        if self.direct:
            return
        if self.sharded:
            # every rank contributes the same number of bytes; rank 0's own tiles never travel (its slot carries a dummy)
            if self.rank == 0:
                send = self.dummy  # (torch.distributed.gather wants a contribution from every rank; pt_gather does not)
            elif self.rgb:
                send = self._send_store[: self.batch * self.own_px]
                self.ops.pack_rgb(self._own_store[b], self.batch * self.own_px, send)
            else:
                send = self.own[b].view(self.batch * self.own_px, 4)
            if getattr(self.ops, "gather_parts", None):
                # the C-ABI's exchange (pt_gather: grouped RCCL send/recv on the render stream); rank r's part lands in gathered[r]
                nbytes = self.batch * self.other_px * (12 if self.rgb else 16)
                self.ops.gather_parts(None if self.rank == 0 else send, self.gathered[1] if self.rank == 0 else None, nbytes)
            else:
                dist.gather(send, self.gather_list if self.rank == 0 else None, dst=0)
        elif self.rehearse and self.rank == 0:
            flat = self.own[b].view(self.batch * self.own_px, 4)
            dist.gather(flat, [self._own_store[b ^ 1][: flat.shape[0]]] if self.world == 1 else None, dst=0)
        if self.rank != 0:
            return
        first, run, stride = self.range
        for f in range(n_frames):
            frame = self.frames[f]
            self.ops.unpack(self._own_store[b], f * self.own_px, 0, 1, first, run, stride, frame)
            if self.sharded:
                # gathered[1:] = ranks 1..N-1, each `batch * other_px` apart; frame f of every part starts f * other_px in
                (self.ops.unpack_rgb if self.rgb else self.ops.unpack)(
                    self._gathered_store, self.batch * self.other_px + f * self.other_px, self.batch * self.other_px,
                    self.world - 1, run, 1, stride, frame)

    # ---- partition tuning ---------------------------------------------------------------------------------------------
    def autotune(self, run_frames, sync, candidates=None, log=None):
        """time `run_frames(exchange)` (a few frames through submit/finish, bracketed by `sync`) for every candidate root
        weight and keep the fastest; the decision is rank 0's (broadcast), from the slowest rank's time per candidate."""
        if self.world == 1:
            return self.root_weight
        if candidates is None:
            candidates = [1, 2, 3, 4, 6, 8, 12, 16, 24, 0]
        times = []
        for k in candidates:
            self.configure(k)
            run_frames(self)  # untimed: buffers, launch-grid estimates and communicators settle
            sync()
            t = run_frames(self)
            tt = torch.tensor([t], dtype=torch.float64, device=self._own_store[0].device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            times.append(float(tt.item()))
        best = torch.tensor([min(range(len(candidates)), key=lambda i: times[i])], dtype=torch.int64, device=self._own_store[0].device)
        dist.broadcast(best, src=0)
        choice = candidates[int(best.item())]
        if log is not None:
            log.update({"candidates": candidates, "seconds": times, "chosen": choice})
        self.configure(choice)
        return choice


class HipOps:
    """ops protocol over the HIP renderer (binding.Renderer): `set_frame(k)` installs frame k's camera and constants.
    cabi_gather=True: the exchange goes through the C-ABI (pt_comm_init has been called on the renderer, pt_gather moves the
    tiles) -- the path a C++ host takes (host/TileExchange.hpp); otherwise through torch.distributed.gather."""

    def __init__(self, renderer, device, set_frame, cabi_gather=False):
        self.r, self.device, self.set_frame = renderer, device, set_frame
        if cabi_gather:
            self.gather_parts = self._gather_parts

    def _gather_parts(self, send, recv, nbytes):
        self.r.gather(send.data_ptr() if send is not None else 0, recv.data_ptr() if recv is not None else 0, nbytes, 0)

    def alloc(self, n_px, channels=4):
        return torch.zeros((max(int(n_px), 1), channels), dtype=torch.float32, device=self.device)

    def pack_rgb(self, src, n_px, dst):
        self.r.pack_rgb(src.data_ptr(), int(n_px), dst.data_ptr())

    def unpack_rgb(self, packed, offset_px, part_stride_px, n_parts, first0, run, stride, frame):
        self.r.unpack_tiles_rgb(packed.data_ptr() + 12 * int(offset_px), int(part_stride_px), n_parts, first0, run, stride, frame.data_ptr())

    def set_range(self, first, run, stride):
        self.r.set_partition_ex(first, run, stride)

    def render(self, k, out):
        self.set_frame(k)
        self.r.render_tiles(out.data_ptr())

    def render_full(self, k, frame):
        self.set_frame(k)
        self.r.render_device(frame.data_ptr())

    def unpack(self, packed, offset_px, part_stride_px, n_parts, first0, run, stride, frame):
        self.r.unpack_tiles_ex(packed.data_ptr() + 16 * int(offset_px), int(part_stride_px), n_parts, first0, run, stride, frame.data_ptr())
