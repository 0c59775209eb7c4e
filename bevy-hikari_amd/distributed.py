"""Band-sharded rendering across GPUs: the frame is cut into horizontal bands of the render image, one per
GPU, with the scene replicated.  Every pass is per-pixel with a bounded screen-space footprint, so the only data that
crosses GPUs are halo rows between neighbouring bands:

    stage TEMPORAL      prepass (+apron), albedo, direct_lit x2, indirect on the band
    exchange A          temporal reservoirs, 20 rows (10 for the emissive channel)   -> spatial_reuse
    stage SPATIAL       spatial_reuse on the band
    exchange B          render (15 rows) + variance (16 rows) per denoised channel   -> denoiser
    stage POST_PROCESS  demodulation + 4 a-trous levels on band + shrinking apron, tone mapping
and, only when the camera or objects moved (history_rows > 0), before stage TEMPORAL:
    exchange C          last frame's temporal + spatial reservoirs, history_rows rows -> reprojection across the border

Which rows of which buffer move, in which order, is decided by the library (`hk_band_plan_for`, `hk_band_schedule`: pure
host logic) and - on GPUs - the transfers themselves run INSIDE the library:

  * one process per GPU: `hk_comm_init` attaches an RCCL communicator to the context and `hk_frame_render` performs the
    exchanges itself on the context's stream (ncclSend / ncclRecv over xGMI).  `BandRenderer(transport="rccl")` only does the
    rendezvous: rank 0's `hk_comm_unique_id` bytes travel through torch.distributed's (gloo) object broadcast.
  * one process, several GPUs: `MultiEngine` = `hk_multi_*` (peer copies ordered by events; what a Bevy render thread drives).

`BandRenderer(transport="host")` is the TEST transport: it executes the same `hk_band_schedule` with torch.distributed
isend / irecv through host memory - over gloo on CPU (the oracle as the compute) and for several ranks sharing ONE GPU (RCCL
refuses two ranks on one device).  Buffers are allocated full-frame on every rank (288 GB HBM makes the 1.6 GB @1080p /
6.6 GB @4K irrelevant), so a halo row lands at the address it has on its owner and no coordinate translation exists anywhere.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


class _DevView:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it zero-copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def halo_plan(width, height, upscale_ratio, band_index, band_count, stage, frame_number, settings_c, bounds=None):
    """Halo transfers band `band_index` must RECEIVE before `stage` (list of HkHaloOp); bounds: an explicit split or None."""
    api = F.api()
    n = F.u32(0)
    b = None if bounds is None else (C.c_uint32 * len(bounds))(*[int(x) for x in bounds])
    api.call("band_plan_bounds", width, height, upscale_ratio, b, band_index, band_count, stage, frame_number, C.byref(settings_c), None, C.byref(n))
    ops = (F.HkHaloOp * max(n.value, 1))()
    n2 = F.u32(n.value)
    if n.value:
        api.call("band_plan_bounds", width, height, upscale_ratio, b, band_index, band_count, stage, frame_number, C.byref(settings_c), ops, C.byref(n2))
    return [ops[i] for i in range(n2.value)]


def band_schedule(width, height, upscale_ratio, rank, n_ranks, stage, frame_number, settings_c, bounds=None):
    """hk_band_schedule(_bounds): every transfer `rank` takes part in before `stage`, sends and receives, in the global order all
    ranks agree on (list of HkTransfer).  bounds: explicit split of the scaled render rows (n_ranks + 1 entries) or None."""
    api = F.api()
    n = F.u32(0)
    b = None if bounds is None else (C.c_uint32 * len(bounds))(*[int(x) for x in bounds])
    api.call("band_schedule_bounds", width, height, upscale_ratio, b, rank, n_ranks, stage, frame_number, C.byref(settings_c), None, C.byref(n))
    tr = (F.HkTransfer * max(n.value, 1))()
    n2 = F.u32(n.value)
    if n.value:
        api.call("band_schedule_bounds", width, height, upscale_ratio, b, rank, n_ranks, stage, frame_number, C.byref(settings_c), tr, C.byref(n2))
    return [tr[i] for i in range(n2.value)]


def _final_buffer(settings, antialias):
    """hk_final_buffer for a library without it (the oracle behind the same table in the CPU tests)."""
    if not antialias:
        return F.BUF_TONE_MAPPED
    if settings.upscale.kind == F.UPSCALE_FSR1:
        return F.BUF_UPSCALE_SHARPENED
    return F.BUF_TAA_OUTPUT if settings.taa == F.TAA_JASMINE else F.BUF_UPSCALE_OUTPUT


def band_gather_schedule(width, height, upscale_ratio, upscale_kind, rank, n_ranks, root, buffer, bounds=None):
    """hk_band_gather_schedule: the transfers of `rank` when band `root` collects every band's rows of `buffer` (list of HkTransfer)."""
    api = F.api()
    b = None if bounds is None else (C.c_uint32 * len(bounds))(*[int(x) for x in bounds])
    n = F.u32(0)
    api.call("band_gather_schedule", width, height, upscale_ratio, upscale_kind, b, rank, n_ranks, root, buffer, None, C.byref(n))
    tr = (F.HkTransfer * max(n.value, 1))()
    n2 = F.u32(n.value)
    if n.value:
        api.call("band_gather_schedule", width, height, upscale_ratio, upscale_kind, b, rank, n_ranks, root, buffer, tr, C.byref(n2))
    return [tr[i] for i in range(n2.value)]


def history_rows_bound(view, previous_view, render_rows, scene_min, scene_max, moved=()):
    """hk_history_rows_bound (pure host logic): the history halo a frame with these uniforms needs over a scene with these world
    bounds; moved: HkMovedBox records of the instances whose model changed since the last frame."""
    n = F.u32(0)
    mn, mx = (F.f32 * 3)(*scene_min), (F.f32 * 3)(*scene_max)
    arr = (F.HkMovedBox * max(len(moved), 1))(*moved)
    F.api().call("history_rows_bound", C.byref(view), C.byref(previous_view), int(render_rows), mn, mx, arr if moved else None, len(moved), C.byref(n))
    return n.value


def balanced_band_bounds(row_costs, width, render_rows, band_count, min_rows=8, background_cost=0.0):
    """hk_balanced_band_bounds: boundaries (band_count + 1 scaled render rows) that give every band about the same cost
    (geometry pixels + width x background_cost per row; 0 = 1/16)."""
    rc = np.ascontiguousarray(row_costs, dtype=np.uint32)
    out = (C.c_uint32 * (band_count + 1))()
    F.api().call("balanced_band_bounds", rc.ctypes.data_as(C.POINTER(C.c_uint32)), len(rc), width, render_rows, band_count, min_rows, background_cost, out)
    return [int(x) for x in out]


def rebalanced_band_bounds(bounds, band_ms, render_rows, row_weight=None, min_rows=8, max_shift=0, damping=0.5):
    """hk_rebalanced_band_bounds (pure host logic): the split `bounds` moved `damping` of the way towards equal MEASURED band times
    `band_ms` (one number per band; a band's time spread over its rows evenly or by the prior row_weight[render_rows]), at most
    max_shift rows per boundary (0: no limit), at least min_rows per band.  bounds None = the equal split."""
    n = len(band_ms)
    if bounds is None:
        base, rem = divmod(int(render_rows), n)
        bounds = [i * base + min(i, rem) for i in range(n)] + [int(render_rows)]
    b = (C.c_uint32 * (n + 1))(*[int(x) for x in bounds])
    ms = (C.c_float * n)(*[float(x) for x in band_ms])
    w = None
    if row_weight is not None:
        rw = np.ascontiguousarray(row_weight, dtype=np.float32)
        assert len(rw) == int(render_rows)
        w = rw.ctypes.data_as(C.POINTER(C.c_float))
    out = (C.c_uint32 * (n + 1))()
    F.api().call("rebalanced_band_bounds", b, ms, n, int(render_rows), w, int(min(min_rows, max(1, int(render_rows) // n))), int(max_shift), float(damping), out)
    return [int(x) for x in out]


def band_migration_schedule(width, height, upscale_ratio, old_bounds, new_bounds, rank, n_ranks, next_frame_number, settings_c):
    """hk_band_migration_schedule: the transfers of `rank` when the split changes from old_bounds to new_bounds before frame
    `next_frame_number` - the rows of the history reservoirs that change owner (list of HkTransfer; None = the equal split)."""
    api = F.api()
    arr = lambda b: None if b is None else (C.c_uint32 * len(b))(*[int(x) for x in b])
    ob, nb = arr(old_bounds), arr(new_bounds)
    n = F.u32(0)
    api.call("band_migration_schedule", width, height, upscale_ratio, ob, nb, rank, n_ranks, int(next_frame_number), C.byref(settings_c), None, C.byref(n))
    tr = (F.HkTransfer * max(n.value, 1))()
    n2 = F.u32(n.value)
    if n.value:
        api.call("band_migration_schedule", width, height, upscale_ratio, ob, nb, rank, n_ranks, int(next_frame_number), C.byref(settings_c), tr, C.byref(n2))
    return [tr[i] for i in range(n2.value)]


class BandRenderer:
    """Drives one rank's band of the frame; `engine` is a bevy_hikari_amd.Engine (or, in the CPU tests, the oracle behind
    the same class).  transport: "rccl" (the product: exchanges inside the library) or "host" (tests, see the module text)."""

    def __init__(self, engine, rank, world_size, backend_device="cuda", transport=None, fallback=None, bounds=None):
        """fallback: what to do when transport "rccl" cannot come up on every rank.  None (default) raises the same
        RuntimeError on ALL ranks - a job that asked for RCCL never silently becomes a PCIe-through-host job; "host" agrees
        on the host-staged transport instead (slower, same bytes) and says so in `transport`."""
        import torch

        self.torch = torch
        self.engine, self.rank, self.world = engine, rank, world_size
        self.device = backend_device
        self.transport = transport or ("host" if backend_device == "cpu" else "rccl")
        self._views = {}
        self._plans = {}
        self._generation = getattr(engine, "generation", 0)
        self.rccl_error = None
        self.bounds = None
        self._band_ms = 0.0
        engine.set_band(rank, world_size)
        if bounds is not None:
            self.set_bounds(bounds)
        if self.transport == "rccl" and world_size > 1:
            import torch.distributed as dist

            # ncclCommInitRank blocks until every rank has called it, so the ranks AGREE before anyone enters it
            # (ADVICE r02): 1. every rank reports hk_comm_available (librccl loads, the device can be made current)
            # without raising; 2. all-reduce MIN over the (gloo) process group; 3. only if all passed, rank 0 creates the
            # id, its 128 bytes travel by object broadcast, and every rank calls hk_comm_init; 4. a second all-reduce
            # collects the outcome of the init itself.
            ok, why = engine.comm_available() if hasattr(engine, "comm_available") else (False, "engine has no RCCL entry points")
            if self._all_ok(ok):
                box = [None]
                if rank == 0:
                    try:
                        box = [engine.comm_unique_id()]
                    except Exception as err:  # (HikariError from the library)
                        why = repr(err)
                dist.broadcast_object_list(box, src=0)
                if box[0] is None:
                    ok, why = False, why or "rank 0 could not create an RCCL id"
                else:
                    try:
                        engine.comm_init(rank, world_size, box[0])
                    except Exception as err:
                        ok, why = False, repr(err)
                    if not self._all_ok(ok) and ok:
                        engine.comm_destroy()
                        ok, why = False, "hk_comm_init failed on another rank"
            else:
                ok, why = False, why or "hk_comm_available failed on another rank"
            if not ok:
                self.rccl_error = why
                if fallback != "host":
                    raise RuntimeError(f"rank {rank}: the RCCL halo transport did not come up on every rank ({why}); "
                                       "pass fallback='host' to stage the halos through host memory instead")
                self.transport = "host (rccl unavailable on some rank: " + why + ")"

    def _all_ok(self, ok):
        import torch.distributed as dist

        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def set_bounds(self, bounds):
        """Bands of unequal height (hk_set_band_bounds): EVERY rank must set the same boundaries.  None = the equal split."""
        self.bounds = None if bounds is None else [int(b) for b in bounds]
        self.engine.set_band_bounds(self.bounds)
        self._plans = {}

    def _follow_resize(self):
        """Engine.resize frees every buffer, bumps engine.generation and - on the C side - drops an explicit band split (it was in
        rows of the old render image): views, plans AND the split of an older generation go (a stale split here would
        move the wrong halo rows while hk_frame_stage renders the equal split)."""
        gen = getattr(self.engine, "generation", 0)
        if gen != self._generation:
            self._views, self._plans, self._generation = {}, {}, gen
            self.bounds = None

    # ------------------------------------------------------------------ host transport (tests)
    def _view(self, buf, parity=0):
        # the double-buffered ids (HkBuffer: position, velocity, tone-mapped, TAA) name a different plane on odd and
        # even frames, so views are kept per frame parity; they are taken after hk_frame_begin of such a frame.
        # Engine.resize frees every buffer and bumps engine.generation: views and plans of an older generation are dropped.
        self._follow_resize()
        key = (buf, parity)
        if key not in self._views:
            torch = self.torch
            ptr, _logical = self.engine.device_ptr(buf)
            _w, _h, _bpp = self.engine.buffer_info(buf)
            nbytes = self.engine.allocated_bytes(buf)   # the allocation, not the logical size of the current upscale kind
            if self.device == "cuda":
                t = torch.as_tensor(_DevView(ptr, nbytes), device="cuda")
            else:
                t = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))
            self._views[key] = t
        return self._views[key]

    def invalidate_views(self):  # kept for callers of the round-1 interface; resize is tracked through engine.generation
        self._views = {}
        self._plans = {}

    def _transfers(self, stage, frame_number, settings_c, width, height, upscale_ratio):
        """(is_recv, view slice, peer) of hk_band_schedule.  The schedule depends on the frame number only through its
        parity (reservoir ping-pong), so it is built once per (stage, parity, settings) and reused."""
        self._follow_resize()
        key = (stage, frame_number & 1, width, height, upscale_ratio, bytes(settings_c))
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        out = []
        for t in band_schedule(width, height, upscale_ratio, self.rank, self.world, stage, frame_number, settings_c, self.bounds):
            out.append((bool(t.is_recv), self._view(t.buffer, frame_number & 1)[t.offset:t.offset + t.bytes], t.peer))
        self._plans[key] = out
        return out

    def exchange(self, stage, frame_number, settings_c, width, height, upscale_ratio):
        """Execute the schedule of `stage` through host memory (the engine has been waited for)."""
        if self.world == 1:
            return 0
        import torch.distributed as dist

        staged = self.device == "cuda"
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
            self.torch.cuda.synchronize()  # the halo rows are in place before the next stage is enqueued on the engine's stream
        return nbytes

    def rebalance(self, my_band_ms, next_frame_number, settings, width, height, damping=0.5, max_shift=0, min_rows=8, row_weight=None):
        """Bands of equal MEASURED time (round 6): every rank contributes the time its band took (`band_time_ms()` after a frame rendered
        with time_band=True, or any per-band figure), ONE all-gather of world_size floats over the process group, every rank evaluates
        hk_rebalanced_band_bounds on the same numbers, the rows of the history reservoirs that change owner travel (RCCL inside the
        library, or the host transport in the tests), and the new split is in force for frame `next_frame_number`.  Returns the new
        boundaries (unchanged boundaries: nothing moves).  Call it between two frames, on every rank."""
        import torch.distributed as dist

        if self.world == 1:
            return self.bounds
        self._follow_resize()
        t = self.torch.zeros(self.world, dtype=self.torch.float32)
        t[self.rank] = float(my_band_ms)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)   # (an all-gather of one float per rank)
        _w, rh, _b = self.engine.buffer_info(F.BUF_TONE_MAPPED)
        new = rebalanced_band_bounds(self.bounds, [float(x) for x in t], rh, row_weight, min_rows, max_shift, damping)
        return self.migrate(new, next_frame_number, settings, width, height)

    def migrate(self, new_bounds, next_frame_number, settings, width, height):
        """The split changes to `new_bounds` (the same on every rank) before frame `next_frame_number`: the rows of the history
        reservoirs that change owner travel to their new owners (hk_migrate_bands over RCCL, or the host transport in the tests), then
        the new split is in force.  Returns the boundaries."""
        import torch.distributed as dist

        self._follow_resize()
        new = [int(b) for b in new_bounds]
        old = self.bounds
        if self.world == 1 or (old is not None and new == [int(b) for b in old]):
            return self.bounds
        sc = settings.to_c()
        if self.transport == "rccl":
            self.engine.migrate_bands(new, next_frame_number, sc)
            self.bounds = new
            self._plans = {}
            return new
        # host transport (tests): the migration's transfers through host memory, then the new split
        self.engine.wait()
        ops, landing = [], []
        for tr in band_migration_schedule(width, height, settings.upscale.ratio(), old, new, self.rank, self.world, next_frame_number, sc):
            view = self._view(tr.buffer, next_frame_number & 1)[tr.offset:tr.offset + tr.bytes]
            if tr.is_recv:
                if self.device == "cuda":
                    tmp = self.torch.empty(view.numel(), dtype=self.torch.uint8)
                    landing.append((view, tmp))
                    view = tmp
                ops.append(dist.P2POp(dist.irecv, view, tr.peer))
            else:
                ops.append(dist.P2POp(dist.isend, view.cpu() if self.device == "cuda" else view.clone(), tr.peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, tmp in landing:
            view.copy_(tmp)
        if landing:
            self.torch.cuda.synchronize()
        self.set_bounds(new)
        return new

    def band_time_ms(self):
        """the band's own time in the last frame rendered with time_band=True (without the waits for the neighbours' halos)"""
        if self.transport == "rccl":
            return self.engine.band_time_ms()
        return self._band_ms

    def gather(self, buffer, settings, width, height, frame_number, root=0):
        """SURVEY 8e step 7 through the host transport (tests): band `root` collects every band's rows of `buffer`."""
        import torch.distributed as dist

        if self.world == 1:
            return
        self.engine.wait()
        ops, landing = [], []
        for t in band_gather_schedule(width, height, settings.upscale.ratio(), settings.upscale.kind, self.rank, self.world, root, buffer, self.bounds):
            view = self._view(t.buffer, frame_number & 1)[t.offset:t.offset + t.bytes]
            if t.is_recv:
                if self.device == "cuda":
                    tmp = self.torch.empty(view.numel(), dtype=self.torch.uint8)
                    landing.append((view, tmp))
                    view = tmp
                ops.append(dist.P2POp(dist.irecv, view, t.peer))
            else:
                ops.append(dist.P2POp(dist.isend, view.cpu() if self.device == "cuda" else view, t.peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for view, tmp in landing:
            view.copy_(tmp)
        if landing:
            self.torch.cuda.synchronize()

    def render(self, frame, view, previous_view, lights, settings, width, height, history_rows=None, antialias=False, balance=False, gather=False, time_band=False):
        """One frame of this rank's band.  history_rows: rows of last frame's reservoirs fetched from the neighbouring bands before
        stage TEMPORAL (exchange C; the bands then also hand each other the scatter stores that cross a border, with exchange A).
        None = derived per frame by the library from the frame's own uniforms (hk_history_rows_bound: 0 for a static view), a
        number overrides it.  balance: split THIS frame's rows by
        cost first (HK_FRAME_BALANCE_BANDS / hk_balance_bands - every rank derives the same split from its own full-frame primary
        rays) and keep the split; meant for the first frame or a cut (rows that change owner lose their history)."""
        e = self.engine
        sc = settings.to_c()
        self._follow_resize()
        if self.world > 1:
            if history_rows is None and "history_rows_bound" not in e.api._fns:
                # (an engine whose library has no host logic of its own - the CPU tests' checker behind the same class: the bound
                # comes from the product library's pure function, for the static scenes those tests render)
                _w, rh, _b = e.buffer_info(F.BUF_TONE_MAPPED)
                history_rows = history_rows_bound(view, previous_view, rh, *e.scene_bounds())
            e.set_history_rows(F.HISTORY_AUTO if history_rows is None else int(history_rows))
        if self.transport == "rccl":
            e.frame_render(frame, view, previous_view, lights, sc,
                           (F.FRAME_ANTIALIAS if antialias else 0) | (F.FRAME_BALANCE_BANDS if balance else 0) | (F.FRAME_GATHER if gather else 0) |
                           (F.FRAME_TIME_BAND if time_band else 0))
            if balance:
                self.bounds = e.band_bounds()
            return
        ratio = settings.upscale.ratio()
        e.frame_begin(frame, view, previous_view, lights)
        history_rows = e.history_rows() if self.world > 1 else 0   # what the library settled on (every rank: the same number)
        if balance and self.world > 1:
            e.set_view_options(sc.taa, sc.upscale_kind, sc.upscale_sharpness)   # (the primary rays' sub-pixel jitter follows the settings)
            self.bounds = e.balance_bands()
            self._plans = {}
        if history_rows > 0:
            e.wait()
            self.exchange(F.STAGE_TEMPORAL | (int(history_rows) << 8), frame.number, sc, width, height, ratio)
        import time as _time

        t0 = _time.perf_counter()
        e.frame_stage(F.STAGE_TEMPORAL, sc)
        e.wait()
        t1 = _time.perf_counter()
        self.exchange(F.STAGE_SPATIAL | (int(history_rows) << 8), frame.number, sc, width, height, ratio)
        t2 = _time.perf_counter()
        e.frame_stage(F.STAGE_SPATIAL, sc)
        e.wait()
        self._band_ms = ((t1 - t0) + (_time.perf_counter() - t2)) * 1e3   # (the host transport's band time: the two stages, not the exchanges)
        self.exchange(F.STAGE_POST_PROCESS, frame.number, sc, width, height, ratio)
        e.frame_stage(F.STAGE_POST_PROCESS, sc)
        if antialias:  # SMAA Tu4x / TAA on the band: exchange D = tone-mapped rows + last frame's TAA rows
            e.wait()
            self.exchange(F.STAGE_ANTIALIAS | (int(history_rows) << 8), frame.number, sc, width, height, ratio)
            e.frame_stage(F.STAGE_ANTIALIAS, sc)
            if settings.upscale.kind == F.UPSCALE_FSR1:  # FSR1 on the band's window rows: exchange E = the EASU taps' input rows
                e.wait()
                self.exchange(F.STAGE_UPSCALE, frame.number, sc, width, height, ratio)
                e.frame_stage(F.STAGE_UPSCALE, sc)
        if gather:   # rank 0 collects the finished image
            self.gather(e.api.final_buffer(sc, F.FRAME_ANTIALIAS if antialias else 0) if hasattr(e.api, "final_buffer") else _final_buffer(settings, antialias),
                        settings, width, height, frame.number)

    def band(self, rows):
        """Rows [b0, b1) of a plane of `rows` rows this rank owns (the render rows; other heights are cut where the boundaries fall)."""
        self._follow_resize()
        if self.bounds is not None:
            rr = self.bounds[-1]
            cut = lambda k: 0 if k == 0 else (rows if k == self.world else (self.bounds[k] if rows == rr else self.bounds[k] * rows // rr))
            return cut(self.rank), cut(self.rank + 1)
        base, rem = divmod(rows, self.world)
        b0 = self.rank * base + min(self.rank, rem)
        return b0, b0 + base + (1 if self.rank < rem else 0)


class MultiEngine:
    """hk_multi_*: one process, n GPUs, one band per context; the library moves the halo rows itself (peer copies ordered by
    events).  device_ids may repeat - several bands on one GPU - which is how the path is tested on a one-GPU box."""

    def __init__(self, device_ids, flags=0):
        from .plugin import Engine

        self.api = F.api()
        self.h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        self.api.call("multi_create", len(device_ids), ids, flags | F.DEFAULT_CTX_FLAGS, C.byref(self.h))
        self.n = len(device_ids)
        self.contexts = []
        for i in range(self.n):
            c = C.c_void_p()
            self.api.call("multi_context", self.h, i, C.byref(c))
            self.contexts.append(Engine.borrowed(self.api, c))

    def close(self):
        if getattr(self, "h", None):
            self.api.raw("multi_destroy")(self.h)
            self.h = None
            for e in self.contexts:  # the contexts died with the hk_multi: a later call through a borrowed Engine must not reach the library
                e.ctx = C.c_void_p()
            self.contexts = []

    def __del__(self):
        self.close()

    def upload_scene(self, scene):
        for e in self.contexts:
            e.upload_scene(scene)   # (textures included)

    def upload_noise(self, noise=None):
        for e in self.contexts:
            e.upload_noise(noise)

    def refit_instances(self, builder):
        """hk_multi_refit_scene_instances: the poses set on `builder` go to every band's device copy of the scene."""
        moved = C.c_uint32()
        self.api.call("multi_refit_scene_instances", self.h, builder.h, C.byref(moved))
        return moved.value

    def rebuild_trees(self, mode=F.TREE_SAH):
        self.api.call("multi_rebuild_scene_trees", self.h, mode)

    def update_instances_on_device(self, builder, mode=F.TREE_SAH):
        self.api.call("multi_update_scene_instances", self.h, builder.h, mode)

    def gather(self, buffer, root=0):
        """hk_multi_gather: band `root`'s context collects every band's rows of `buffer` on its own device."""
        self.api.call("multi_gather", self.h, buffer, root)

    def migrate_bands(self, new_bounds, next_frame_number, settings_c):
        """hk_multi_migrate_bands: the history rows that change owner travel between the bands' contexts (peer copies), then every
        band takes the new split (None = equal)."""
        if new_bounds is None:
            self.api.call("multi_migrate_bands", self.h, None, 0, int(next_frame_number), C.byref(settings_c))
        else:
            arr = (C.c_uint32 * len(new_bounds))(*[int(b) for b in new_bounds])
            self.api.call("multi_migrate_bands", self.h, arr, len(new_bounds), int(next_frame_number), C.byref(settings_c))

    def set_band_bounds(self, bounds=None):
        """hk_multi_set_band_bounds: bands of unequal height, the same split on every context (None = equal)."""
        if bounds is None:
            self.api.call("multi_set_band_bounds", self.h, None, 0)
        else:
            arr = (C.c_uint32 * len(bounds))(*[int(b) for b in bounds])
            self.api.call("multi_set_band_bounds", self.h, arr, len(bounds))

    def resize(self, width, height, upscale_ratio=1.0):
        self.api.call("multi_resize", self.h, width, height, upscale_ratio)
        for e in self.contexts:
            e.generation += 1

    def set_history_rows(self, rows):
        self.api.call("multi_set_history_rows", self.h, rows)

    def frame_render(self, frame, view, previous_view, lights, settings_c, flags=0):
        self.api.call("multi_frame_render", self.h, C.byref(frame), C.byref(view), C.byref(previous_view), C.byref(lights), C.byref(settings_c), flags)

    def wait(self):
        self.api.call("multi_wait", self.h)

    def read(self, buf):
        """The union of the bands' own rows of `buf` (same array shape as Engine.read)."""
        e = self.contexts[0]
        w, h, bpp = e.buffer_info(buf)
        raw = np.empty(w * h * bpp, dtype=np.uint8)
        self.api.call("multi_read_buffer", self.h, buf, raw.ctypes.data_as(C.c_void_p), raw.size)
        return e.shape_buffer(buf, raw)
