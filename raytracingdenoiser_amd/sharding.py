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
