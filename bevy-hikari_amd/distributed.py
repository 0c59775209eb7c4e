"""Band-sharded rendering across GPUs: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The frame is cut into horizontal bands of the render image, one per rank, with the scene
replicated.  Every pass is per-pixel with a bounded screen-space footprint, so the only
data that crosses ranks are halo rows, exchanged point-to-point between neighbouring bands twice
per frame:

    stage TEMPORAL      prepass (+apron), albedo, direct_lit x2, indirect on the band
    exchange A          temporal reservoirs, 20 rows (10 for the emissive channel)   -> spatial_reuse
    stage SPATIAL       spatial_reuse on the band
    exchange B          render (15 rows) + variance (16 rows) per denoised channel   -> denoiser
    stage POST_PROCESS  demodulation + 4 a-trous levels on band + shrinking apron, tone mapping
and, only when the camera or objects moved (history_rows > 0), before stage TEMPORAL:
    exchange C          last frame's temporal + spatial reservoirs, history_rows rows -> reprojection across the border

Which rows of which buffer move is decided by the library (`hk_band_plan_for`, pure host logic);
this module only executes that plan with isend/irecv on zero-copy views of the library's device
buffers.  Buffers are allocated full-frame on every rank (288 GB HBM makes the 1.6 GB @1080p /
6.6 GB @4K irrelevant), so a halo row lands at the same address it has on its owner and no
coordinate translation exists anywhere.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


class _DevView:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it zero-copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def halo_plan(width, height, upscale_ratio, band_index, band_count, stage, frame_number, settings_c):
    """Halo transfers band `band_index` must RECEIVE before `stage` (list of HkHaloOp)."""
    api = F.api()
    n = F.u32(0)
    api.call("band_plan_for", width, height, upscale_ratio, band_index, band_count, stage, frame_number, C.byref(settings_c), None, C.byref(n))
    ops = (F.HkHaloOp * max(n.value, 1))()
    n2 = F.u32(n.value)
    if n.value:
        api.call("band_plan_for", width, height, upscale_ratio, band_index, band_count, stage, frame_number, C.byref(settings_c), ops, C.byref(n2))
    return [ops[i] for i in range(n2.value)]


class BandRenderer:
    """Drives one rank's band of the frame; `engine` is a bevy_hikari_amd.Engine (or, in the CPU
    tests, the oracle behind the same class)."""

    def __init__(self, engine, rank, world_size, backend_device="cuda"):
        import torch

        self.torch = torch
        self.engine, self.rank, self.world = engine, rank, world_size
        self.device = backend_device
        self._views = {}
        self._plans = {}
        engine.set_band(rank, world_size)

    def _view(self, buf, parity=0):
        # the double-buffered ids (HkBuffer: position, velocity, tone-mapped, TAA) name a different plane on odd and
        # even frames, so views are kept per frame parity; they are taken after hk_frame_begin of such a frame
        key = (buf, parity)
        if key not in self._views:
            torch = self.torch
            ptr, nbytes = self.engine.device_ptr(buf)
            if self.device == "cuda":
                t = torch.as_tensor(_DevView(ptr, nbytes), device="cuda")
            else:
                t = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))
            self._views[key] = t
        return self._views[key]

    def invalidate_views(self):  # after hk_resize
        self._views = {}
        self._plans = {}

    def _transfers(self, stage, frame_number, settings_c, width, height, upscale_ratio):
        """(is_recv, view slice, peer) for every transfer of `stage` this rank takes part in, in a
        global order all ranks agree on.  The plan depends on the frame number only through its
        parity (reservoir ping-pong), so it is built once per (stage, parity, settings) and reused:
        at 8 GPUs a band's frame is a few hundred microseconds and per-frame host work would
        otherwise dominate."""
        key = (stage, frame_number & 1, width, height, upscale_ratio, bytes(settings_c))
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        out = []
        for peer_rank in range(self.world):  # fixed global order: plans of rank 0, 1, ...
            for op in halo_plan(width, height, upscale_ratio, peer_rank, self.world, stage, frame_number, settings_c):
                lo, hi = op.row_begin * op.row_bytes, op.row_end * op.row_bytes
                if peer_rank == self.rank:       # I receive rows owned by op.peer
                    out.append((True, self._view(op.buffer, frame_number & 1)[lo:hi], op.peer))
                elif op.peer == self.rank:       # peer_rank needs rows I own
                    out.append((False, self._view(op.buffer, frame_number & 1)[lo:hi], peer_rank))
        self._plans[key] = out
        return out

    def exchange(self, stage, frame_number, settings_c, width, height, upscale_ratio):
        """Execute the halo plan of `stage` for every rank pair this rank takes part in."""
        if self.world == 1:
            return 0
        import torch.distributed as dist

        # RCCL moves device memory directly; gloo (CPU tests, and the one-GPU multi-rank test) cannot
        # address device memory, so device views are staged through host tensors there.
        staged = self.device == "cuda" and dist.get_backend() != "nccl"
        ops, nbytes, landing = [], 0, []
        for is_recv, view, peer in self._transfers(stage, frame_number, settings_c, width, height, upscale_ratio):
            if is_recv:
                if staged:
                    tmp = self.torch.empty(view.numel(), dtype=self.torch.uint8)
                    landing.append((view, tmp))
                    view = tmp
                ops.append(dist.P2POp(dist.irecv, view, peer))
                nbytes += view.numel()
            else:
                ops.append(dist.P2POp(dist.isend, view.cpu() if staged else view, peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, tmp in landing:
            view.copy_(tmp)
        if landing:
            self.torch.cuda.synchronize()
        return nbytes

    def render(self, frame, view, previous_view, lights, settings, width, height, history_rows=0, antialias=False):
        """One frame: three stages with the two halo exchanges in between.  history_rows > 0 (camera or
        objects moved since the last frame) first fetches that many rows of last frame's reservoirs from the
        neighbouring bands (exchange C, HK_STAGE_TEMPORAL_WITH_HISTORY): reprojection may cross the band border."""
        e = self.engine
        sc = settings.to_c()
        ratio = settings.upscale.ratio()
        e.frame_begin(frame, view, previous_view, lights)
        if history_rows > 0:
            self._sync_before_exchange()
            self.exchange(F.STAGE_TEMPORAL | (int(history_rows) << 8), frame.number, sc, width, height, ratio)
        e.frame_stage(F.STAGE_TEMPORAL, sc)
        self._sync_before_exchange()
        self.exchange(F.STAGE_SPATIAL, frame.number, sc, width, height, ratio)
        e.frame_stage(F.STAGE_SPATIAL, sc)
        self._sync_before_exchange()
        self.exchange(F.STAGE_POST_PROCESS, frame.number, sc, width, height, ratio)
        e.frame_stage(F.STAGE_POST_PROCESS, sc)
        if antialias:  # SMAA Tu4x / TAA on the band: exchange D = tone-mapped rows + last frame's TAA rows
            self._sync_before_exchange()
            self.exchange(F.STAGE_ANTIALIAS | (int(history_rows) << 8), frame.number, sc, width, height, ratio)
            e.frame_stage(F.STAGE_ANTIALIAS, sc)
            if settings.upscale.kind == F.UPSCALE_FSR1:  # FSR1 on the band's window rows: exchange E = the EASU taps' input rows
                self._sync_before_exchange()
                self.exchange(F.STAGE_UPSCALE, frame.number, sc, width, height, ratio)
                e.frame_stage(F.STAGE_UPSCALE, sc)

    def _sync_before_exchange(self):
        # When the engine runs on torch's current stream (Engine.set_stream), RCCL orders itself
        # against that stream and nothing is needed.  On the engine's own stream (or the CPU
        # oracle) wait for the stage to finish first.
        if not getattr(self.engine, "on_host_stream", False):
            self.engine.wait()

    def band(self, rows):
        base, rem = divmod(rows, self.world)
        b0 = self.rank * base + min(self.rank, rem)
        return b0, b0 + base + (1 if self.rank < rem else 0)
