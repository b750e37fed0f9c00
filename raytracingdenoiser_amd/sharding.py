"""Multi-GPU frame sharding: one process per GPU, torch.distributed over RCCL/xGMI (gloo on CPU for the tests).

Scheme (DESIGN.md "Multi-GPU"): the frame is cut into `world` horizontal row strips. Every rank keeps FULL-SIZE planes
(258 MB at 1440p, nothing against 288 GB) but its executor only produces its own strip, each pass being launched on the strip
extended by the reach of the later passes (include/NRDHip.h: nrdHipSetOwnedRows), so the owned rows are bit-identical to a
single-GPU frame. After the last pass ONE collective phase -- an in-place all-gather per plane, owned strips being contiguous
row ranges of the pitched planes -- completes the outputs and the permanent (history) planes on every rank, which is what the
next frame's temporal reprojection needs. If the height is not divisible by the world size the sharder falls back to replicas
(every rank denoises the whole frame, no collective).
"""


def strip_rows(height, rank, world):
    """Owned row range of `rank`, or None when the frame cannot be cut evenly (replica fallback)."""
    if world <= 1 or height % world != 0:
        return None
    rows = height // world
    return rank * rows, (rank + 1) * rows


def exchange_strips(planes, row_begin, row_end, group=None):
    """In-place all-gather of rows [row_begin, row_end) of every plane (2D uint8 tensors [H, pitch], identical layout on all ranks)."""
    import torch.distributed as dist

    for p in planes:
        assert p.dim() == 2 and p.is_contiguous()

    def gather_all():
        for p in planes:
            dist.all_gather_into_tensor(p.view(-1), p[row_begin:row_end].reshape(-1), group=group)

    # RCCL: issue the per-plane all-gathers as ONE grouped collective (a dozen small launches per frame otherwise); gloo (CPU tests)
    # and anything unexpected take the plain loop
    if planes and planes[0].is_cuda and dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager"):
        try:
            with dist._coalescing_manager(group=group, device=planes[0].device, async_ops=False):
                gather_all()
            return
        except Exception:  # noqa: BLE001 -- private torch API: fall back rather than fail the frame
            try:  # drop a half-recorded op list, or the plain calls below would be recorded instead of executed
                c10d = dist.distributed_c10d
                c10d._world.pg_coalesce_state.pop(group or c10d._get_default_group(), None)
            except Exception:  # noqa: BLE001
                pass
    gather_all()


class FrameSharder:
    def __init__(self, executor, instance, width, height, rank, world, outputs, group=None):
        from . import api

        self.ex, self.inst = executor, instance
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        self.rows = strip_rows(height, rank, world)
        self.planes = []
        if self.rows is not None:
            executor.set_owned_rows(*self.rows)
            for i, (fmt, downsample) in enumerate(instance.permanent_pool):
                if downsample == 1:
                    self.planes.append(executor.pool_plane_tensor(api.ResourceType.PERMANENT_POOL, i))
            for t in outputs:  # user outputs: [H, W, C] tensors -> byte rows
                assert t.is_contiguous() and t.shape[0] == height
                self.planes.append(t.view(-1).view(dtype=__import__("torch").uint8).view(height, -1))
            # The full-resolution transient planes too (round 6, "texels nobody writes" -- DESIGN.md section 6): every pass skips the sky, so a transient plane keeps in its sky texels
            # what the LAST writer of an earlier frame left there, and passes with a neighbourhood read those texels (RELAX TemporalAccumulation: the pre-pass output's hit distances).
            # A rank's last writer covers only strip + its own margin; reassembling the planes makes their unwritten texels those of one GPU everywhere.
            for i, (fmt, downsample) in enumerate(instance.transient_pool):
                if downsample == 1:
                    self.planes.append(executor.pool_plane_tensor(api.ResourceType.TRANSIENT_POOL, i))

    def pixels_per_rank(self):
        """Pixels of the OWNED strip (the algorithmic work of a rank; halo rows are redundant work, not useful output)."""
        if self.rows is None:
            return self.width * self.height
        return self.width * (self.rows[1] - self.rows[0])

    def exchange(self):
        if self.rows is not None:
            exchange_strips(self.planes, self.rows[0], self.rows[1], self.group)

    def denoise(self):
        self.ex.denoise()
        self.exchange()


# ======================================================================================================================
# Halo-exchange sharding: instead of recomputing the halo of every later pass (FrameSharder: up to ~110 extra rows per side at the
# default radii, i.e. 2-4x redundant work on 180-row strips), the dispatch list is cut into SEGMENTS in front of every pass with a
# large vertical reach (blur / post-blur / a-trous at large steps ...). Inside a segment a rank still recomputes the small halos
# (temporal accumulation +-1, history fix ...); at a segment boundary it receives from its two neighbours the boundary band of every
# plane the next passes read -- rows the neighbours own and have just produced -- with one grouped batch of point-to-point transfers
# (RCCL send / recv over the xGMI links to the two neighbouring ranks; gloo in the CPU tests). The planes carried over from the
# previous frame (histories, previous guides) are exchanged the same way at the start of the frame, widened by the largest vertical
# motion the application promises (max_motion_rows). Everything is planned from the DispatchDesc list itself (which planes a pass
# reads / writes, nrdHipGetDispatchReach for its reach), so any denoiser the executor can bound is covered; a frame with a pass of
# unknown reach (the clears of a restart frame, hit-distance reconstruction) simply runs unsharded on every rank, which leaves all
# planes complete everywhere. Results stay bit-identical to one GPU as long as motion < max_motion_rows - 2 (HaloSharder checks the camera part every frame).
class HaloPlan:
    def __init__(self):
        self.fallback = False      # run the whole frame on every rank, no exchange
        self.steps = []            # [(exchange_items, first, count)] with exchange_items = [(plane_key, width_rows)]
        self.row_begin, self.row_end = [], []
        self.margins, self.reach = [], []
        self.early = []            # per step: leading dispatches that do not touch the exchanged planes (run during the transfers)
        self.complete_keys = []
        self.output_keys = []      # the user's OUT_* planes this frame writes: all-gathered after the last pass (HaloSharder.gather_outputs)
        self._c_rows = None        # ctypes copies of row_begin / row_end (made once: plans are cached and reused every other frame)
        self._ops = {}             # step -> (plane addresses, P2POp list, bytes received): the transfer batch, rebuilt only if a plane moved

    def c_rows(self):
        import ctypes as C

        if self._c_rows is None:
            n = len(self.row_begin)
            self._c_rows = ((C.c_int32 * n)(*self.row_begin), (C.c_int32 * n)(*self.row_end))
        return self._c_rows


def _is_user_input(resource_type):
    from . import api

    return api.ResourceType(resource_type).name.startswith("IN_")


def plan_halo_exchange(dispatches, reach, rows, height, max_motion_rows=32, exchange_threshold=24, small_planes=(), min_strip=None):
    """dispatches: api.Dispatch list of one frame; reach: nrdHipGetDispatchReach; rows: (row_begin, row_end) owned by this rank;
    small_planes: keys of the down-sampled pool planes (tile maps): the passes writing them run on the whole frame on every rank (they
    are tiny and only read user inputs), so these planes are complete everywhere and never exchanged."""
    from . import api

    plan = HaloPlan()
    n = len(dispatches)
    plan.reach = list(reach)
    if rows is None or n == 0 or any(r < 0 for r in reach):
        plan.fallback = True
        return plan
    rb, re = rows
    per = min_strip if min_strip is not None else re - rb  # a halo must fit into the neighbouring strip
    starts = [0] + [i for i in range(1, n) if reach[i] > exchange_threshold]
    bounds = starts + [n]
    seg_of = [0] * n
    margins = [0] * n
    for s in range(len(starts)):
        a, b = bounds[s], bounds[s + 1]
        m = 0
        for i in range(b - 1, a - 1, -1):
            seg_of[i] = s
            margins[i] = m
            m += reach[i]
    plan.margins = margins

    last_write = {}
    cleared = set()  # planes a Clear_ pass of this frame wrote (every texel, on every rank)
    whole_frame = set()
    need = {}  # (key, writer index or -1) -> halo rows
    for i, d in enumerate(dispatches):
        reads = [(int(t), idx) for dt, t, idx in d.resources if dt == api.DescriptorType.TEXTURE and not _is_user_input(t) and (int(t), idx) not in small_planes]
        writes = [(int(t), idx) for dt, t, idx in d.resources if dt == api.DescriptorType.STORAGE_TEXTURE and not _is_user_input(t)]
        if any(key in small_planes for key in writes):
            whole_frame.add(i)
            if any(key not in small_planes for key in writes) or reads:
                plan.fallback = True  # a tile-map pass that also touches full-resolution pool planes: not expected, stay safe
                return plan
            continue
        for key in reads:
            w = last_write.get(key, -1)
            # a plane written by SIGMA's Copy IS last frame's history (previous output -> HISTORY, texel to texel): TemporalStabilization samples it at the reprojected
            # position, so its readers need the motion bound like readers of a carried-over plane (executor.hip nrdHipPlanHaloExchange: the same rule)
            history_copy = w >= 0 and dispatches[w].shader.startswith("SIGMA_Copy")
            if history_copy and seg_of[w] == seg_of[i]:
                plan.fallback = True
                return plan
            if w >= 0 and seg_of[w] == seg_of[i]:
                # produced in this segment with a sufficient margin -- where it was WRITTEN. Every pass skips the sky, and a reader with a neighbourhood (reach > 0) also reads the
                # texels next to the geometry that the writer left alone (RELAX TemporalAccumulation takes the specular hit distance of its 3x3 from the pre-pass output, sky or
                # not: RELAX_TemporalAccumulation.hlsli:363, 433-450). On one GPU such a texel holds what the plane held at the end of the last frame; on a rank that is only true
                # inside its strip. So the rows of the reader's neighbourhood that lie outside the strip are fetched from their owner at the frame start, BEFORE the writer runs
                # (round 6: found by the 4-rank 4K model run, whose second strip starts just below the horizon).
                if reach[i] > 0 and key not in cleared:  # (a plane cleared earlier in this frame holds zeros wherever nothing was written since: the same on every rank)
                    need[(key, -1)] = max(need.get((key, -1), 0), margins[i] + reach[i])
                continue
            h = margins[i] + reach[i] + (max_motion_rows if w < 0 or history_copy else 0)
            if h > 0:
                need[(key, w)] = max(need.get((key, w), 0), h)
        for key in writes:
            last_write[key] = i
            if d.shader.startswith("Clear_"):
                cleared.add(key)
    if any(h > per for h in need.values()):
        plan.fallback = True  # a halo would reach past the neighbouring strip
        return plan

    exchanges = [[] for _ in starts]
    for (key, w), h in sorted(need.items()):
        exchanges[0 if w < 0 else seg_of[w] + 1].append((key, h))
    for s in range(len(starts)):
        plan.steps.append((exchanges[s], bounds[s], bounds[s + 1] - bounds[s]))
        # leading dispatches of the segment that neither read nor write a plane of this exchange: they can run while the transfers are in flight
        keys = {key for key, _ in exchanges[s]}
        early = 0
        for i in range(bounds[s], bounds[s + 1]):
            if any((int(t), idx) in keys for dt, t, idx in dispatches[i].resources):
                break
            early += 1
        plan.early.append(early if keys else 0)
    plan.row_begin = [-1 if i in whole_frame else max(rb - margins[i], 0) for i in range(n)]
    plan.row_end = [height if i in whole_frame else min(re + margins[i], height) for i in range(n)]
    return plan


def halo_transfers(rows, rank, world, items):
    """[(plane_index, 'send' | 'recv', peer, row_begin, row_end)] for one exchange; items = [(plane_index, width_rows)]."""
    rb, re = rows
    ops = []
    for k, w in items:
        if rank > 0:
            ops.append((k, "send", rank - 1, rb, rb + w))
            ops.append((k, "recv", rank - 1, rb - w, rb))
        if rank < world - 1:
            ops.append((k, "send", rank + 1, re - w, re))
            ops.append((k, "recv", rank + 1, re, re + w))
    return ops


def start_halo_exchange(planes, rows, rank, world, items, group=None):
    """planes: list of 2D uint8 tensors [H, pitch]; items: [(index into planes, width)]. Issues one grouped batch of sends / receives and
    returns the pending requests: over RCCL the transfers then run on the communicator's stream, next to whatever the caller launches on the
    compute stream until it calls finish_halo_exchange (which only makes the compute stream wait -- the host does not block)."""
    import torch.distributed as dist

    transfers = halo_transfers(rows, rank, world, items)
    if not transfers:
        return None
    staged = planes[0].is_cuda and dist.get_backend(group) != "nccl"  # gloo moves host memory: stage the bands (tests: two ranks sharing one GPU)
    ops, landing = [], []
    for k, kind, peer, r0, r1 in transfers:
        band = planes[k][r0:r1]
        if staged:
            host = band.cpu() if kind == "send" else __import__("torch").empty(band.shape, dtype=band.dtype)
            if kind == "recv":
                landing.append((band, host))
            band = host
        ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, band, global_rank(group, peer), group))
    return dist.batch_isend_irecv(ops), landing


def finish_halo_exchange(pending):
    if pending is None:
        return
    requests, landing = pending
    for req in requests:
        req.wait()
    for band, host in landing:
        band.copy_(host)


def exchange_halos(planes, rows, rank, world, items, group=None):
    finish_halo_exchange(start_halo_exchange(planes, rows, rank, world, items, group))


def carried_over_planes(dispatches, small_planes=(), reach=None):
    """keys of the planes an UNSHARDED frame needs complete on every rank before it runs: those it reads before (or without) writing them -- the history it inherits from the previous
    frame -- and, given the per-dispatch reach, those a pass with a neighbourhood (reach != 0; -1 = unbounded) reads AFTER an earlier pass of the frame wrote them: every pass skips the
    sky, so such a plane keeps in its sky texels what the last frame left there ("texels nobody writes"), and on a rank that is only complete inside its own strip.
    (Round 6: found by a dynamic-resolution step -- an unsharded frame in which silhouettes move by whole pixels -- with 3 uniform strips.)"""
    from . import api

    written, carried = set(), []
    for i, d in enumerate(dispatches):
        neighbourhood = reach is not None and reach[i] != 0
        for dt, t, idx in d.resources:
            key = (int(t), idx)
            if dt == api.DescriptorType.TEXTURE and not _is_user_input(t) and key not in small_planes and (key not in written or neighbourhood) and key not in carried:
                carried.append(key)
        for dt, t, idx in d.resources:
            if dt == api.DescriptorType.STORAGE_TEXTURE:
                written.add((int(t), idx))
    return carried


def global_rank(group, rank_in_group):
    """torch.distributed addresses point-to-point peers and broadcast sources by GLOBAL rank; strips are numbered by the rank inside the sharder's group (the two
    only coincide for the default group)"""
    import torch.distributed as dist

    return rank_in_group if group is None or group is dist.group.WORLD else dist.get_global_rank(group, rank_in_group)


def output_planes_of(dispatches):
    """keys of the user's OUT_* planes the list writes"""
    from . import api

    keys = []
    for d in dispatches:
        for dt, t, idx in d.resources:
            if api.ResourceType(t).name.startswith("OUT_") and dt == api.DescriptorType.STORAGE_TEXTURE and (int(t), idx) not in keys:
                keys.append((int(t), idx))
    return keys


def output_gather_ops(bounds, keys):
    """the reassembly of the output planes as data movement: [(plane key, source rank, row_begin, row_end)] -- every rank ends up with every other rank's rows
    (one all-gather per plane; the virtual-rank tests replay the list with copies)"""
    return [(key, src, bounds[src], bounds[src + 1]) for key in keys for src in range(len(bounds) - 1) if bounds[src + 1] > bounds[src]]


def balanced_bounds(tile_row_cost, height, world, min_rows, tile=16, max_rows=None):
    """Strip boundaries (rows, multiples of `tile` except the last) that minimise the largest strip cost; every strip is at least min_rows
    high (and at most max_rows, if given). tile_row_cost[t] = cost of tile row t. Returns None when the frame is too small for `world` such strips."""
    T = len(tile_row_cost)
    min_tiles = max(1, -(-min_rows // tile))
    max_tiles = T if max_rows is None else max(min_tiles, max_rows // tile)
    if T < world * min_tiles or T > world * max_tiles:
        return None
    prefix = [0.0]
    for c in tile_row_cost:
        prefix.append(prefix[-1] + float(c))
    INF = float("inf")
    best = [[INF] * (T + 1) for _ in range(world + 1)]
    cut = [[0] * (T + 1) for _ in range(world + 1)]
    best[0][0] = 0.0
    for k in range(1, world + 1):
        for t in range(k * min_tiles, T - (world - k) * min_tiles + 1):
            for s in range(max((k - 1) * min_tiles, t - max_tiles), t - min_tiles + 1):
                if best[k - 1][s] == INF:
                    continue
                v = max(best[k - 1][s], prefix[t] - prefix[s])
                if v < best[k][t]:
                    best[k][t], cut[k][t] = v, s
    if best[world][T] == INF:
        return None
    cuts, t = [T], T
    for k in range(world, 0, -1):
        t = cut[k][t]
        cuts.append(t)
    cuts.reverse()
    return [min(c * tile, height) for c in cuts[:-1]] + [height]


_motion_cache = {}


def camera_motion_rows(cs, depth_range=(1.0, 1.0e4), grid=5):
    """Estimate of the vertical reprojection distance (in rows) of STATIC surface points between the previous and the current camera of a
    CommonSettings: a grid of screen positions x the two ends of a view-depth range is un-projected with the current matrices and re-projected with the
    previous ones (rotation moves all depths alike, translation moves the nearest depth most: depth_range[0] must not exceed the view depth of the
    nearest geometry, an application-level fact -- HaloSharder(near_depth=...)). A heuristic on a 5 x 5 sample, not a proven upper bound: the caller
    doubles it and adds a margin. Returns None for an orthographic projection, infinity when no bound can be given (a singular view matrix, points
    behind the previous camera). One matrix product for all samples; the result is cached on the matrix bytes (a static camera costs a dict lookup)."""
    import numpy as np

    raw = bytes(cs.viewToClipMatrix) + bytes(cs.viewToClipMatrixPrev) + bytes(cs.worldToViewMatrix) + bytes(cs.worldToViewMatrixPrev)
    key = (raw, float(cs.rectSize[1]), tuple(depth_range), grid)
    hit = _motion_cache.get(key)
    if hit is not None:
        return hit[0]

    def mat(m):
        return np.array(list(m), dtype=np.float64).reshape(4, 4).T  # column-major in, M @ v out

    def remember(value):
        if len(_motion_cache) > 64:
            _motion_cache.clear()
        _motion_cache[key] = (value,)
        return value

    P, Pp, V, Vp = mat(cs.viewToClipMatrix), mat(cs.viewToClipMatrixPrev), mat(cs.worldToViewMatrix), mat(cs.worldToViewMatrixPrev)
    if abs(P[3, 2]) < 1e-12 or abs(P[0, 0]) < 1e-12 or abs(P[1, 1]) < 1e-12:
        return remember(None)
    try:
        Vinv = np.linalg.inv(V)
    except np.linalg.LinAlgError:
        return remember(float("inf"))  # singular / unset view matrix: treated as exceeding any halo
    if not np.all(np.isfinite(Vinv)):
        return remember(float("inf"))
    sign = 1.0 if P[3, 2] > 0 else -1.0  # clip.w = +z (left-handed) or -z (right-handed)
    h = float(cs.rectSize[1])
    n = np.linspace(-1.0, 1.0, grid)
    nx, ny, zv = [a.ravel() for a in np.meshgrid(n, n, sign * np.asarray(depth_range, dtype=np.float64), indexing="ij")]
    w = P[3, 2] * zv + P[3, 3]
    view = np.stack([(nx * w - P[0, 2] * zv - P[0, 3]) / P[0, 0], (ny * w - P[1, 2] * zv - P[1, 3]) / P[1, 1], zv, np.ones_like(zv)])
    clip = (Pp @ Vp @ Vinv) @ view
    if np.any(clip[3] <= 1e-9):
        return remember(float("inf"))  # behind the previous camera: no bound
    return remember(float(np.max(np.abs(clip[1] / clip[3] - ny)) * 0.5 * h))


class HaloSharder:
    """Row-strip sharding with halo exchange between pass segments (see above). `denoise()` replaces executor.denoise().

    Motion contract: the history halos cover reprojection over fewer than max_motion_rows rows (the Catmull-Rom footprint needs motion + 2 rows, so
    motion == max_motion_rows is already too much). Every frame the vertical camera motion of static geometry is estimated from the matrices of the
    last CommonSettings (camera_motion_rows), doubled for the virtual motion of specular reflections, plus whatever bound the application passes for
    moving objects (denoise(motion_rows=...)); a frame that exceeds the halo runs UNSHARDED on every rank (after completing the planes) instead of
    silently reading stale rows. All ranks take the same decision from the same settings.
    measure_motion=True replaces the estimate AND the application's promise by a measurement: every rank reduces IN_VIEWZ / IN_MV of its own strip on the
    device (nrdHipMeasureMotionRows: the temporal passes' own surface-motion reprojection, moving objects included), the ranks take the MAX (one 4-byte
    all-reduce per frame), and the virtual-motion factor is applied to that.

    balance: after a frame that ran unsharded (the restart frame, a dynamic-resolution step, ...) every rank holds every plane completely, so
    the strips can be re-cut for free: they are then chosen from the tile map so that every rank gets the same number of non-sky tiles
    (sky tiles cost almost nothing), not the same number of rows."""

    SKY_TILE_COST = 0.03  # relative to a tile with geometry (early-out blocks still pay their launch and the tile test)

    @staticmethod
    def default_motion_rows(height):
        """history-halo width when the caller names none: 32 rows up to 1440p, growing with the frame height above it (motion in rows scales with the resolution:
        with a fixed 32 the 4K bench sequence exceeded the halo on every frame and ran unsharded, profiles/r03_i_scaling_model_relax_ds_sh.json)"""
        return max(32, -(-height * 32 // 1440))

    def __init__(self, executor, instance, width, height, rank, world, group=None, max_motion_rows=None, exchange_threshold=24, balance=True, recut_every=0, near_depth=1.0,
                 measure_motion=False, gather_outputs=True):
        max_motion_rows = self.default_motion_rows(height) if max_motion_rows is None else max_motion_rows
        self.measure_motion, self.measured_motion_rows = measure_motion, None  # the last measurement (rows), for reporting
        self.near_depth = near_depth  # view depth of the nearest geometry the camera-motion estimate has to cover (scene units)
        self.ex, self.inst = executor, instance
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        self.max_motion_rows, self.exchange_threshold, self.balance = max_motion_rows, exchange_threshold, balance
        # recut_every = N > 0: every N sharded frames one frame is run unsharded on purpose (after completing the planes), so that the strips
        # are re-cut from a fresh tile map and follow the camera; costs one whole-frame time + one plane completion per N frames
        self.recut_every, self._sharded_since_cut = recut_every, 0
        # point-to-point halos do not need equal strips: any height splits (the all-gather scheme needs height % world == 0)
        self.bounds = [r * height // world for r in range(world + 1)] if world > 1 and height >= world else None
        self.complete = True      # every plane is complete on this rank: fresh (zeroed) arena, or the last frame ran unsharded
        self.exchanged_bytes = 0  # received bytes, for reporting
        self.rebalanced = 0
        self.motion_fallbacks = 0  # frames run unsharded because the motion estimate exceeded the history halo
        # History reach (round 6): what the measured SURFACE motion cannot bound -- the specular passes' virtual-motion position and the look-back taps behind it -- is REPORTED by
        # the temporal kernels themselves (include/NRDHip.h nrdHipSetHistoryReachWord: rows, MAX over the rank's pixels). With measure_motion the word rides on the same per-frame
        # all-reduce as the motion measurement: frame f is decided with max(2 x motion_f + 2, 1.25 x reach_{f-1} + 3), and a frame whose own reach turns out to have left the halo
        # it ran with is COUNTED (history_halo_violations; the next frame then falls back by the rule above, so stale rows are read for one frame at most -- and never unnoticed).
        self.history_reach_rows = 0.0     # the last value known (the previous frame's, over all ranks)
        self.history_halo_violations = 0  # sharded frames whose temporal passes read beyond the halo they were given
        self._last_frame_sharded = False
        self._words = None                # [motion of this frame, history reach of the previous one]: float32 x 2 on the executor's device
        self._plans = {}
        # Output reassembly (BASELINE.json configs[3]: "screen tiled across 8 x MI355X with RCCL all-gather"; the reference's Integration::Denoise hands back COMPLETE outputs,
        # NRDIntegration.hpp:516-623). The bound OUT_* planes are working planes of the pass chain (the pre-pass of the NEXT frame already overwrites them), so the complete
        # planes are separate tensors owned by the sharder (complete_output): after the last pass a rank copies its rows into them (device-to-device, microseconds) and the ranks
        # all-gather them in place -- asynchronously over RCCL, on the communicator's stream behind an event of the compute stream -- while the whole next frame runs.
        # wait_outputs() is what a consumer of the outputs calls before complete_output().
        self.gather_outputs = gather_outputs
        self._pending_gather = None
        self._complete = {}
        self.gathered_bytes = 0   # bytes received by this rank in output all-gathers, for reporting
        self.gather_frames = 0

    def close(self):
        """unregisters the history-reach word (it lives in a tensor of this object; the executor may outlive it)"""
        if self._words is not None and hasattr(self.ex, "set_history_reach_word"):
            try:
                self.ex.set_history_reach_word(None)
            except Exception:
                pass  # (the executor is gone already)
        self._words = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self):
        return (self.bounds[self.rank], self.bounds[self.rank + 1]) if self.bounds else None

    @property
    def min_strip(self):
        return min(b - a for a, b in zip(self.bounds, self.bounds[1:])) if self.bounds else self.height

    def pixels_per_rank(self):
        if self.rows is None:
            return self.width * self.height
        return self.width * (self.rows[1] - self.rows[0])

    def plane_tensor(self, key):
        """2D uint8 view [H, pitch] of a pool plane or of a bound user plane"""
        from . import api

        t, idx = key
        if t in (int(api.ResourceType.PERMANENT_POOL), int(api.ResourceType.TRANSIENT_POOL)):
            return self.ex.pool_plane_tensor(api.ResourceType(t), idx)
        bound = self.ex._bound[t]
        assert bound.is_contiguous(), "halo exchange needs densely packed user planes"
        return bound.view(-1).view(dtype=__import__("torch").uint8).view(self.height, -1)

    def _small_planes(self):
        from . import api

        return {(int(pool), i) for pool, descs in ((api.ResourceType.PERMANENT_POOL, self.inst.permanent_pool), (api.ResourceType.TRANSIENT_POOL, self.inst.transient_pool))
                for i, (fmt, downsample) in enumerate(descs) if downsample != 1}

    def _tile_row_cost(self):
        """per 16-row band: tiles with geometry + a little for the sky tiles, from the tile map the previous frame left in the pool (R8: 255 = sky)"""
        from . import api

        for i, (fmt, downsample) in enumerate(self.inst.transient_pool):
            if downsample == 16 and fmt == api.Format.R8_UNORM:
                tiles = self.ex.pool_plane_tensor(api.ResourceType.TRANSIENT_POOL, i)[:, : (self.width + 15) // 16].cpu().numpy()
                return [float((row == 0).sum()) + self.SKY_TILE_COST * len(row) for row in tiles]
        return None

    SPECULAR_MOTION_FACTOR = 2.0  # virtual (reflection) motion relative to the surface motion that the camera estimate bounds

    HISTORY_REACH_MARGIN, HISTORY_FOOTPRINT_ROWS = 1.25, 3.0  # frame f's reach is predicted from frame f - 1's; the bicubic footprint around a sample position

    def _measure_motion_over_ranks(self, dispatches, rows):
        """MAX over ranks of (this frame's surface motion on the rank's own rows, the history reach the temporal kernels reported LAST frame). Over RCCL the values never visit
        the host before they are reduced (round 6): the kernels write into device words, the all-reduce reads them there in stream order, and the host
        synchronises ONCE, on the reduced pair -- until round 5: stream-synchronise, 4-byte read-back, upload, all-reduce, read-back. Over gloo (the tests: CPU emulation, or two
        ranks sharing one GPU) the pair is staged through the host. Returns the motion; the reach lands in self.history_reach_rows."""
        import torch
        import torch.distributed as dist

        real_group = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) == self.world
        if self.world > 1 and not real_group:
            # virtual ranks of a single process: nobody reduces for them, and a rank deciding on its LOCAL maximum could take a different fallback decision than its
            # neighbours -- the host that drives virtual ranks has to reduce the measurements itself (begin_frame(motion_rows=..., history_reach=...))
            raise RuntimeError("HaloSharder(measure_motion=True) with %d ranks but no process group of that size: reduce the measured motion over the ranks yourself "
                               "(ex.measure_motion_rows per rank, then denoise(motion_rows=max)) or create the sharder with measure_motion=False" % self.world)
        on_device = hasattr(self.ex, "measure_motion_rows_async") and getattr(self.ex, "arena", None) is not None and self.ex.arena.is_cuda
        if self._words is None:
            self._words = torch.zeros(2, dtype=torch.float32, device="cuda" if on_device else "cpu")
            if hasattr(self.ex, "set_history_reach_word"):
                self.ex.set_history_reach_word(self._words[1:])
        if on_device:
            self.ex.measure_motion_rows_async(dispatches[0], dispatches[1], rows[0], rows[1], self._words[:1])
        else:
            self._words[0] = self.ex.measure_motion_rows(dispatches[0], dispatches[1], rows[0], rows[1])
        if self.world > 1:
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(self._words, dist.ReduceOp.MAX, self.group)
                motion, reach = self._words.tolist()  # the one synchronisation of the frame's planning
            else:
                host = self._words.cpu()
                dist.all_reduce(host, dist.ReduceOp.MAX, self.group)
                motion, reach = host.tolist()
        else:
            motion, reach = self._words.tolist()
        self._words[1:].zero_()  # this frame's temporal passes report into a cleared word (stream order: behind the reduction, in front of the frame's launches)
        self.history_reach_rows = float(reach)
        if self._last_frame_sharded and reach + self.HISTORY_FOOTPRINT_ROWS > self.max_motion_rows:
            self.history_halo_violations += 1
        return float(motion)

    def motion_exceeds_halo(self, motion_rows=None, dispatches=None, history_reach=None):
        """True when this frame's reprojection may leave the history halo (see the class docstring); dispatches = (ptr, n) of this frame's list (measure_motion);
        history_reach: hosts that reduce over (virtual) ranks themselves pass what the temporal kernels reported LAST frame, MAX over the ranks (nrdHipSetHistoryReachWord)"""
        if history_reach is not None:
            self.history_reach_rows = float(history_reach)
            if self.HISTORY_REACH_MARGIN * self.history_reach_rows + self.HISTORY_FOOTPRINT_ROWS >= self.max_motion_rows:
                return True
        if getattr(self, "measure_motion", False) and dispatches is not None:
            rows = self.rows or (0, self.height)
            self.measured_motion_rows = self._measure_motion_over_ranks(dispatches, rows)
            need = max(self.SPECULAR_MOTION_FACTOR * self.measured_motion_rows + 2.0, self.HISTORY_REACH_MARGIN * self.history_reach_rows + self.HISTORY_FOOTPRINT_ROWS)
            return need >= self.max_motion_rows
        cs = getattr(self.inst, "last_common_settings", None)
        camera = camera_motion_rows(cs, (getattr(self, "near_depth", 1.0), 1.0e4)) if cs is not None else 0.0
        if camera is None:
            camera = 0.0  # orthographic: rejected by the launchers anyway
        need = self.SPECULAR_MOTION_FACTOR * camera + (float(motion_rows) if motion_rows is not None else 0.0) + 2.0
        return need >= self.max_motion_rows

    def begin_frame(self, motion_rows=None, history_reach=None):
        """GetComputeDispatches + plan; returns (plan, dispatch pointer, count). A plan that falls back while this rank's planes are incomplete
        carries plan.complete_keys: the carried-over planes every rank has to receive in full before the frame runs."""
        from . import api

        import ctypes as C

        r, ptr, n = self.inst.get_compute_dispatches_raw()
        assert r == api.Result.SUCCESS, r
        reach = self.inst.dispatch_reach(ptr, n)
        # The plan depends on which planes each pass binds (ping-pong: period 2) and on the reach, not on the per-frame constants: steady-state
        # frames hit this cache and skip the Python-side parsing and planning (~0.3 ms, as much as a strip's GPU time at 8 ranks).
        signature = (tuple(self.bounds) if self.bounds else None, tuple(reach),
                     tuple((ptr[i].pipelineIndex, C.string_at(C.addressof(ptr[i].resources.contents), ptr[i].resourcesNum * C.sizeof(api.ResourceDesc)) if ptr[i].resourcesNum else b"")
                           for i in range(n)))
        cached = self._plans.get(signature)
        recut = self.balance and self.recut_every > 0 and self._sharded_since_cut >= self.recut_every and self.world > 1
        if self.balance and self.world > 1 and self.complete and any(self.inst.pipelines[ptr[i].pipelineIndex].startswith("Clear_") for i in range(n) if ptr[i].pipelineIndex < len(self.inst.pipelines)):
            # a restart frame (its list clears the history) with balancing on: run it whole on every rank -- nothing has to be completed for it, and its tile map is what the
            # strips are cut from in front of the next frame. (Without balancing the restart frame is sharded like any other since round 6: clears are texel-local.)
            recut = True
        if self.world > 1 and self.motion_exceeds_halo(motion_rows, (ptr, n), history_reach):
            recut = True  # same mechanics as a deliberate re-cut frame: complete the planes, run the whole frame everywhere
            self.motion_fallbacks += 1
        if cached is not None and not cached.fallback and not self.complete and not recut:
            return cached, ptr, n
        dispatches = [api.Dispatch(ptr[i], self.inst.pipelines) for i in range(n)]
        small = self._small_planes()

        def make_plan():
            return plan_halo_exchange(dispatches, reach, self.rows, self.height, self.max_motion_rows, self.exchange_threshold, small, self.min_strip)

        plan = make_plan()
        if recut and not plan.fallback:
            plan = HaloPlan()
            plan.fallback, plan.reach = True, list(reach)
        if not plan.fallback and self.balance and self.complete and self.world > 1:
            widest = max((w for items, _, _ in plan.steps for _, w in items), default=0)
            cost = self._tile_row_cost()
            # with the output all-gather a strip is also a MESSAGE: every rank sends its rows of the OUT_* planes to each peer over that peer's link, so the tallest strip
            # bounds the frame rate -- a sky strip is cheap to compute and as expensive to send as any other. Strips are therefore capped at 1.5 x the uniform height
            # (1440p, 8 ranks, MODELLED at 50 GB/s per link: 544-row sky strip 0.45 ms per frame on its links against 0.22 ms of compute; capped at 270 rows: 0.22 ms)
            cap = (3 * self.height // (2 * self.world) + 15) & ~15 if self.gather_outputs else None
            new = balanced_bounds(cost, self.height, self.world, max(widest, 16), max_rows=cap) if cost else None
            new = self._agree(new)
            if new and new != self.bounds:
                old, self.bounds = self.bounds, new
                replanned = make_plan()
                if replanned.fallback:
                    self.bounds = old
                else:
                    plan = replanned
                    self.rebalanced += 1
        plan.complete_keys = carried_over_planes(dispatches, small, reach) if plan.fallback and not self.complete and self.world > 1 else []
        plan.output_keys = output_planes_of(dispatches)
        if plan.complete_keys and self.gather_outputs:
            # the OUT_* planes too: texels the frame does not write (a pixel that turned into sky) keep what the LAST frame that wrote them left there -- on a single GPU that
            # is the previous frame's value, on a rank that did not own the row it would be a value from before the strips were cut. Nothing reads those texels, but the
            # reassembled planes are held bit for bit against a single GPU, unwritten texels included.
            plan.complete_keys = plan.complete_keys + [k for k in plan.output_keys if k not in plan.complete_keys]
        if not plan.fallback:
            if len(self._plans) > 16:
                self._plans.clear()
            self._plans[(tuple(self.bounds) if self.bounds else None,) + signature[1:]] = plan  # keyed by the strips actually used (they may just have been re-cut)
        return plan, ptr, n

    def _agree(self, bounds):
        """Every rank derives the same strips from the same tile map; rank 0's answer is nevertheless broadcast (a few bytes, only on the frame
        after an unsharded one) so that a rank that somehow disagreed cannot desynchronise the transfer schedule."""
        import torch
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) != self.world:
            return bounds  # virtual ranks of the single-process tests
        device = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor(bounds if bounds else [-1] * (self.world + 1), dtype=torch.int64, device=device)
        dist.broadcast(t, global_rank(self.group, 0), self.group)
        agreed = [int(v) for v in t.tolist()]
        return agreed if agreed[0] == 0 else None

    def finish_frame(self, plan):
        self.complete = plan.fallback or self.world == 1
        self._last_frame_sharded = not plan.fallback and self.world > 1
        self._sharded_since_cut = 0 if plan.fallback else self._sharded_since_cut + 1

    def run_step(self, plan, ptr, n, step):
        _, first, count = plan.steps[step]
        self.ex.execute_range(ptr, n, first, count, *plan.c_rows())

    def start_exchange(self, plan, step):
        items = plan.steps[step][0]
        if not items or self.world == 1:
            return None
        import torch.distributed as dist

        planes = [self.plane_tensor(key) for key, _ in items]
        where = tuple(p.data_ptr() for p in planes)
        cached = plan._ops.get(step)
        if cached is not None and cached[0] == where:  # same planes at the same addresses as last time: reissue the same batch
            self.exchanged_bytes += cached[2]
            return dist.batch_isend_irecv(cached[1]), []
        pairs = [(k, min(w, planes[k].shape[0])) for k, (_, w) in enumerate(items)]
        received = sum(planes[k].shape[1] * w * ((self.rank > 0) + (self.rank < self.world - 1)) for k, w in pairs)
        self.exchanged_bytes += received
        if planes[0].is_cuda and dist.get_backend(self.group) != "nccl":
            return start_halo_exchange(planes, self.rows, self.rank, self.world, pairs, self.group)  # host-staged (tests): nothing to reuse
        ops = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, planes[k][r0:r1], global_rank(self.group, peer), self.group)
               for k, kind, peer, r0, r1 in halo_transfers(self.rows, self.rank, self.world, pairs)]
        if not ops:
            return None
        plan._ops[step] = (where, ops, received)
        return dist.batch_isend_irecv(ops), []

    def exchange_step(self, plan, step):
        finish_halo_exchange(self.start_exchange(plan, step))

    def complete_planes(self, keys):
        """every rank receives every other rank's strip of the given planes (before an unsharded frame that follows a sharded one)"""
        import torch
        import torch.distributed as dist

        for key in keys:
            plane = self.plane_tensor(key)
            staged = plane.is_cuda and dist.get_backend(self.group) != "nccl"  # gloo moves host memory (tests: two ranks sharing one GPU)
            for src in range(self.world):
                band = plane[self.bounds[src]:self.bounds[src + 1]]
                if band.shape[0] == 0:
                    continue
                if staged:
                    host = band.cpu() if src == self.rank else torch.empty(band.shape, dtype=band.dtype)
                    dist.broadcast(host, global_rank(self.group, src), self.group)
                    if src != self.rank:
                        band.copy_(host)
                else:
                    dist.broadcast(band, global_rank(self.group, src), self.group)
                if src != self.rank:
                    self.exchanged_bytes += band.numel()

    def complete_output(self, resource_type):
        """the COMPLETE output plane of the last frame (all ranks' rows), as a tensor shaped like the bound OUT_* tensor -- valid after wait_outputs(). The bound OUT_* planes
        themselves are working planes of the pass chain (the next frame's pre-pass overwrites them) and hold this rank's rows only."""
        key = (int(resource_type), 0)
        return self._complete[key] if key in self._complete else self.ex._bound[key[0]]  # (one rank: the bound plane is the complete plane)

    def _complete_plane(self, key):
        """[H, pitch] uint8 view of the complete copy of an output plane (allocated on first use, same shape as the bound tensor)"""
        import torch

        bound = self.ex._bound[key[0]]
        t = self._complete.get(key)
        if t is None or t.shape != bound.shape or t.dtype != bound.dtype or t.device != bound.device:
            t = self._complete[key] = torch.zeros_like(bound)
        return t.view(-1).view(dtype=torch.uint8).view(self.height, -1)

    def stage_outputs(self, plan):
        """this rank's rows of every OUT_* plane (the whole plane after an unsharded frame) -> the complete copies. Device-to-device on the compute stream, behind the last pass."""
        self.wait_outputs()  # the previous frame's all-gather still reads / writes the complete copies (it finished a frame ago; over RCCL this is an event wait, no host block)
        rb, re = (0, self.height) if plan.fallback or self.rows is None else self.rows
        for key in plan.output_keys:
            self._complete_plane(key)[rb:re].copy_(self.plane_tensor(key)[rb:re], non_blocking=True)

    def start_output_gather(self, plan):
        """ONE all-gather per OUT_* plane (grouped into one RCCL launch where torch allows it), in place on the complete copies: equal strips -- every rank's rows already sit at
        their offset (all_gather_into_tensor); strips re-cut by the load balancer are unequal -- the list form, which RCCL runs as a group of broadcasts into the row views.
        Asynchronous over RCCL: the collective runs on the communicator's stream behind an event of the compute stream (the staging copies), next to the WHOLE next frame --
        nothing of the next frame touches the complete copies. Returns the pending work (None: nothing to do / done synchronously)."""
        import torch.distributed as dist

        if not self.gather_outputs or not plan.output_keys:
            return None
        in_group = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) == self.world
        if self.world == 1 and not in_group:
            return None  # a plain single-GPU run: the bound planes are the complete planes
        self.stage_outputs(plan)
        if plan.fallback and self.world > 1:
            return None  # an unsharded frame: every rank computed every row
        if not in_group:
            return None  # virtual ranks of the single-process tests: they replay output_gather_ops() with copies between the ranks' complete planes
        # (a process group of ONE rank runs the collective all the same: the only way this box can exercise the RCCL path -- tests/test_sharding.py)
        bounds = self.bounds or [0, self.height]
        planes = [self._complete_plane(key) for key in plan.output_keys]
        rb, re = bounds[self.rank], bounds[self.rank + 1]
        heights = [b - a for a, b in zip(bounds, bounds[1:])]
        self.gathered_bytes += sum(p.shape[1] * (p.shape[0] - (re - rb)) for p in planes)
        self.gather_frames += 1
        if dist.get_backend(self.group) != "nccl":  # gloo (CPU tests, or two ranks sharing one GPU with host staging): synchronous
            import torch

            for p in planes:
                for src in range(self.world):
                    band = p[bounds[src]:bounds[src + 1]]
                    if band.shape[0] == 0:
                        continue
                    if p.is_cuda:
                        host = band.cpu() if src == self.rank else torch.empty(band.shape, dtype=band.dtype)
                        dist.broadcast(host, global_rank(self.group, src), self.group)
                        if src != self.rank:
                            band.copy_(host)
                    else:
                        dist.broadcast(band, global_rank(self.group, src), self.group)
            return None
        uniform = len(set(heights)) == 1 and not __import__("os").environ.get("NRD_HIP_GATHER_LIST_FORM")  # (test hook: the list form with equal strips too)

        def issue():
            works = []
            for p in planes:
                if uniform:
                    works.append(dist.all_gather_into_tensor(p.view(-1), p[rb:re].reshape(-1), group=self.group, async_op=True))
                else:
                    works.append(dist.all_gather([p[bounds[r]:bounds[r + 1]] for r in range(self.world)], p[rb:re], group=self.group, async_op=True))
            return works

        if hasattr(dist, "_coalescing_manager") and uniform:
            try:
                with dist._coalescing_manager(group=self.group, device=planes[0].device, async_ops=True) as cm:
                    for p in planes:
                        dist.all_gather_into_tensor(p.view(-1), p[rb:re].reshape(-1), group=self.group)
                return [cm]
            except Exception:  # noqa: BLE001 -- private torch API: fall back to one collective per plane rather than fail the frame
                try:
                    c10d = dist.distributed_c10d
                    c10d._world.pg_coalesce_state.pop(self.group or c10d._get_default_group(), None)
                except Exception:  # noqa: BLE001
                    pass
        return issue()

    def wait_outputs(self):
        """the compute stream waits for the pending output all-gather (no host block over RCCL); complete_output() then holds every rank's rows of the last frame"""
        pending, self._pending_gather = self._pending_gather, None
        if pending:
            for work in pending:
                work.wait()

    def denoise(self, motion_rows=None):
        """motion_rows: the application's bound on the vertical motion of moving objects this frame (rows); camera motion is estimated here"""
        plan, ptr, n = self.begin_frame(motion_rows)
        if plan.fallback:
            self.complete_planes(plan.complete_keys)
            self.ex.execute_range(ptr, n, 0, n)
        else:
            for step, (_, first, count) in enumerate(plan.steps):
                # the passes in front of the first reader of an exchanged plane (tile classification, pre-pass: user inputs only) hide the transfers
                pending = self.start_exchange(plan, step)
                early = plan.early[step] if pending is not None else 0
                if early:
                    self.ex.execute_range(ptr, n, first, early, *plan.c_rows())
                finish_halo_exchange(pending)
                self.ex.execute_range(ptr, n, first + early, count - early, *plan.c_rows())
        self._pending_gather = self.start_output_gather(plan)
        self.finish_frame(plan)
        return plan


def dry_plan(name, width, height, world, overrides=None, max_motion_rows=None, exchange_threshold=24):
    """The per-rank plan of the halo scheme for one steady-state frame WITHOUT any GPU: strips, the pass segments, per exchange step the
    planes and rows each rank receives from its two neighbours, the bytes, and the RCCL operation list (one grouped batch of send / recv per step and
    neighbour pair). Pure host work: the dispatch list comes from nrd::GetComputeDispatches, the reach from nrdHipGetDispatchReach, the planner is the one
    HaloSharder uses. Returns a JSON-able dict; `bench.py --gpus N --dry-plan` prints it."""
    from . import api, scene, synth

    inst = api.Instance([(0, scene.DENOISERS[name][0])])
    # cameras of the bench sequence (the planes themselves are not needed; the SIGMA settings carry the scene's light direction, which comes with its planes)
    frames = [synth.render_frame(32, 18, f, device="cpu", want=("sigma",) if name.startswith("SIGMA") else ()) for f in range(3)]
    settings = scene.denoiser_settings(name, frames[0], overrides)
    assert inst.set_denoiser_settings(0, settings) == api.Result.SUCCESS
    max_motion_rows = HaloSharder.default_motion_rows(height) if max_motion_rows is None else max_motion_rows
    dispatches = reach = None
    for f in range(3):  # frame 0 restarts (clears: unsharded); frames 1, 2 are the two ping-pong phases of the steady state
        assert inst.set_common_settings(scene.common_settings(frames[f]["camera"], frames[max(f - 1, 0)]["camera"], width, height, f)) == api.Result.SUCCESS
        r, ptr, n = inst.get_compute_dispatches_raw()
        assert r == api.Result.SUCCESS
        reach = inst.dispatch_reach(ptr, n)
        dispatches = [api.Dispatch(ptr[i], inst.pipelines) for i in range(n)]
    small = {(int(pool), i) for pool, descs in ((api.ResourceType.PERMANENT_POOL, inst.permanent_pool), (api.ResourceType.TRANSIENT_POOL, inst.transient_pool))
             for i, (fmt, ds) in enumerate(descs) if ds != 1}
    bounds = [r * height // world for r in range(world + 1)]
    min_strip = min(b - a for a, b in zip(bounds, bounds[1:]))

    def row_bytes(key):
        t, idx = key
        if t == int(api.ResourceType.PERMANENT_POOL):
            fmt = inst.permanent_pool[idx][0]
        elif t == int(api.ResourceType.TRANSIENT_POOL):
            fmt = inst.transient_pool[idx][0]
        else:
            fmt = {int(rt): f for rt, _, _, f in scene.output_planes(name, width, height)}.get(t)
        return (width * api.FORMAT_BYTES[fmt] + 255) & ~255 if fmt is not None else 0

    def label(key):
        t, idx = key
        rt = api.ResourceType(t)
        return "%s[%d]" % (rt.name, idx) if "POOL" in rt.name else rt.name

    out = {"denoiser": name, "size": [width, height], "ranks": world, "strips": bounds, "history_halo_rows": max_motion_rows, "passes": [d.shader for d in dispatches], "reach_rows": list(reach),
           "per_rank": []}
    for rank in range(world):
        plan = plan_halo_exchange(dispatches, reach, (bounds[rank], bounds[rank + 1]), height, max_motion_rows, exchange_threshold, small, min_strip)
        entry = {"rank": rank, "rows": [bounds[rank], bounds[rank + 1]], "fallback_unsharded": bool(plan.fallback), "steps": []}
        total = 0
        for si, (items, first, count) in enumerate(plan.steps):
            neighbours = [r for r in (rank - 1, rank + 1) if 0 <= r < world]
            recv = sum(w * row_bytes(key) for key, w in items) * len(neighbours)
            total += recv
            entry["steps"].append({"before_pass": dispatches[first].shader, "passes_in_segment": count, "planes": [{"plane": label(key), "rows_from_each_neighbour": w, "bytes_per_neighbour": w * row_bytes(key)} for key, w in items],
                                   "rccl": ["group { " + ", ".join("recv(%d rows of %d planes from rank %d), send(same to rank %d)" % (max((w for _, w in items), default=0), len(items), nb, nb) for nb in neighbours) + " }"] if items else [],
                                   "received_bytes": recv, "overlaps_with": "the first %d pass(es) of the segment (they do not touch these planes)" % plan.early[si] if si < len(plan.early) and plan.early[si] else None})
        out_keys = output_planes_of(dispatches)
        own = bounds[rank + 1] - bounds[rank]
        gathered = sum(row_bytes(key) * (height - own) for key in out_keys) if not plan.fallback else 0
        entry["output_all_gather"] = {"planes": [label(key) for key in out_keys], "rows_contributed": own, "received_bytes": gathered,
                                      "rccl": "all_gather_into_tensor per plane, one group (equal strips: in place; re-cut strips: the list form)" if out_keys and not plan.fallback else None,
                                      "overlaps_with": "the whole next frame (the complete planes are separate tensors: a rank stages its rows with a device copy behind the last pass)",
                                      "ms_at_7_links_x_50GBps": round(gathered / (7 * 50e9) * 1e3, 3)}
        entry["received_bytes_per_frame"] = total + gathered
        entry["transfer_ms_at_50GBps_per_link"] = round(total / max(len([r for r in (rank - 1, rank + 1) if 0 <= r < world]), 1) / 50e9 * 1e3, 3)
        out["per_rank"].append(entry)
    return out
