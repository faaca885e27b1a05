"""Multi-GPU glue: instances shard across ranks (one process per GPU), every rank culls its own
shard, and two small all-gathers turn the per-rank results into one global draw list description:

  1. all-gather of each rank's visible count (8 bytes per rank);
  2. every rank rebases its per-instance firstTask by the sum of the lower ranks' counts
     (on the device for the GPU path: mcull_rebase_commands reads the gathered counts);
  3. all-gather of the per-instance indirect commands.

The two-phase frame (cull, depth pre-pass, pyramid, cull with occlusion) adds one exchange: every rank
rasterises only its own instances, so the ranks' depth images are merged with an all-reduce (maximum:
reverse-Z, larger = nearer) before each rank builds the pyramid of the whole scene (merge_depth,
sharded_frame).

The reference is single-GPU (Engine.cpp:658-666 picks one device); this has no counterpart there.
torch.distributed is the transport ("nccl" is RCCL on ROCm, "gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist


def shard_instances(n_instances, world, rank, weights=None):
    """Contiguous instance range [begin, end) of `rank`.  With `weights` (meshlets per instance) the
    cut points balance the summed weight instead of the instance count."""
    if weights is None:
        return (n_instances * rank) // world, (n_instances * (rank + 1)) // world
    import numpy as np

    cum = np.concatenate([[0], np.cumsum(np.asarray(weights, dtype=np.int64))])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, (total * r) // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n_instances
    return cuts[rank], cuts[rank + 1]


def gather_draw_list(total_local, cmds_local, rebase, group=None):
    """total_local: int64 tensor [1]; cmds_local: int32 tensor [n_local, 2] = {taskCount, firstTask}
    with rank-local firstTask; rebase(totals_all, rank) must add sum(totals_all[:rank]) to every
    firstTask of cmds_local in place (stream-ordered).  Every rank must hold the same n_local
    (pad the last shard) because all_gather_into_tensor needs equal sizes.
    Returns (totals_all [world], cmds_all [world * n_local, 2])."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    totals_all = torch.empty(world, dtype=torch.int64, device=total_local.device)
    dist.all_gather_into_tensor(totals_all, total_local, group=group)
    rebase(totals_all, rank)
    cmds_all = torch.empty((world * cmds_local.shape[0], 2), dtype=cmds_local.dtype, device=cmds_local.device)
    dist.all_gather_into_tensor(cmds_all, cmds_local, group=group)
    return totals_all, cmds_all


def gather_visible_lists(list_local, totals_all, out=None, group=None, first_instances=None):
    """Optional third step: one global visible list on every rank.  list_local: int32 tensor
    [capacity, 2] whose first totals_all[rank] rows are this rank's (drawID, meshletID) entries;
    totals_all: the gathered counts.  A rank's cull numbers its draws from 0 (its shard's first
    instance), so the drawIDs of different shards overlap: pass first_instances (one value per rank,
    the global index of that rank's instance 0, e.g. shard_instances(...)[0]) and they are added to
    column 0 while the shards are packed, which makes the result the list of the unsharded scene.
    Without it the drawIDs stay as the ranks wrote them (rank-local unless the caller has already made
    them global), and a consumer has to recover an entry's shard from its position and totals_all.
    The shards differ in length, so every rank
    contributes max(totals_all) rows (one host read of the counts) and the padding is dropped when
    the shards are packed at their scanned offsets — the offsets the rebased firstTask values refer
    to.  Returns an int32 tensor [sum(totals_all), 2]; `out`, when given, must hold at least that
    many rows and receives the packed list (its first sum rows are returned).
    Traffic per rank: world * max(totals) * 8 bytes received; over xGMI this is link-bound and costs
    more than the cull itself, so a frame loop that can address per-rank lists should skip it."""
    world = dist.get_world_size(group)
    counts = [int(x) for x in totals_all.cpu().tolist()]
    pad = max(max(counts), 1)
    if list_local.shape[0] < pad:
        raise ValueError("list_local holds %d rows, the largest shard has %d" % (list_local.shape[0], pad))
    staged = torch.empty((world * pad, 2), dtype=list_local.dtype, device=list_local.device)
    dist.all_gather_into_tensor(staged, list_local[:pad].contiguous(), group=group)
    total = sum(counts)
    if out is None:
        out = torch.empty((max(total, 1), 2), dtype=list_local.dtype, device=list_local.device)
    elif out.shape[0] < total:
        raise ValueError("out holds %d rows, %d needed" % (out.shape[0], total))
    if first_instances is not None and len(first_instances) != world:
        raise ValueError("first_instances needs one value per rank (%d), got %d" % (world, len(first_instances)))
    pos = 0
    for r, c in enumerate(counts):
        out[pos:pos + c] = staged[r * pad:r * pad + c]
        if first_instances is not None and c:
            out[pos:pos + c, 0] += int(first_instances[r])
        pos += c
    return out[:total]


_package = None  # the ctypes binding, imported on first use (this module is importable without the library)


def _pkg():
    global _package
    if _package is None:
        import importlib

        _package = importlib.import_module(__package__)
    return _package


def rebase_gathered(gathered, world, rows, out=None):
    """`gathered` (int32 [world * rows, 2]): per rank rows - 1 {taskCount, firstTask} records with rank-local
    firstTask and one record holding the rank's visible count (uint64).  Adds the counts of the lower ranks to
    every firstTask in place (modulo 2^32 like the 32-bit field it is) and returns (totals_all int64 [world],
    cmds_all int32 [world, rows - 1, 2], a view of `gathered`).  On the GPU this is one launch of the
    library (mcull_rebase_gathered_commands) on the current stream; on the CPU (gloo) a few tensor ops.
    `out` (optional, GPU): an int64 [2, world] tensor for the totals and bases, reused by a frame loop."""
    g = gathered.view(world, rows, 2)
    if gathered.is_cuda:
        global _package
        if _package is None:
            import importlib

            _package = importlib.import_module(__package__)
        if out is None:
            out = torch.empty((2, world), dtype=torch.int64, device=gathered.device)
        with torch.cuda.device(gathered.device):  # the library launches on the current device: make it the tensor's
            _package.rebase_gathered_commands(torch.cuda.current_stream(gathered.device).cuda_stream, gathered.data_ptr(), world, rows - 1,
                                              out[0].data_ptr(), out[1].data_ptr())
        return out[0], g[:, : rows - 1, :]
    return rebase_gathered_tensor_ops(gathered, world, rows)


def rebase_gathered_tensor_ops(gathered, world, rows):
    """rebase_gathered spelled with tensor operations (any device)."""
    g = gathered.view(world, rows, 2)
    lo = g[:, rows - 1, 0].to(torch.int64) & 0xFFFFFFFF
    hi = g[:, rows - 1, 1].to(torch.int64) & 0xFFFFFFFF
    totals_all = lo | (hi << 32)
    base = torch.cumsum(totals_all, 0) - totals_all
    cmds_all = g[:, : rows - 1, :]
    cmds_all[:, :, 1] += base.to(torch.int32)[:, None]
    return totals_all, cmds_all


def gather_draw_list_packed(cmds_packed, group=None, gathered=None, out=None):
    """The same global description with ONE collective per frame.  cmds_packed: int32 tensor
    [n_local + 1, 2]: rows 0..n_local-1 are the rank's {taskCount, firstTask} with rank-local
    firstTask, and row n_local holds the rank's visible count as a little-endian uint64 (pass the
    address of that row as mcull_cull_params.total: the cull writes the count straight into it).
    All ranks' buffers are gathered at once and every rank rebases the gathered firstTask values
    itself from the gathered counts (rebase_gathered: no second collective and no host round trip).
    `gathered` ([world * (n_local + 1), 2], same dtype and device) and `out` (see rebase_gathered): buffers a frame
    loop keeps instead of allocating per frame.
    Returns (totals_all int64 [world], cmds_all int32 [world, n_local, 2])."""
    world = dist.get_world_size(group)
    rows = cmds_packed.shape[0]
    if gathered is None:
        gathered = torch.empty((world * rows, 2), dtype=cmds_packed.dtype, device=cmds_packed.device)
    dist.all_gather_into_tensor(gathered, cmds_packed, group=group)
    return rebase_gathered(gathered, world, rows, out)


def create_native_comm(device, group=None):
    """An RCCL communicator of the library's own (mcull_rccl_comm_create) over the ranks of a torch.distributed group:
    rank 0's unique id travels through the group (any backend), then every rank joins.  The communicator is what
    gather_draw_list_native hands to the library; torch.distributed is only the courier of 128 bytes."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [None]
    if rank == 0:
        try:
            box[0] = _pkg().rccl_unique_id()
        except Exception as e:  # noqa: BLE001 - told to every rank below, so that all of them give up together
            box[0] = RuntimeError("rank 0 could not make an RCCL id: %s" % e)
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if isinstance(box[0], Exception):
        raise box[0]
    return _pkg().RcclComm(box[0], world, rank, device)


def gather_draw_list_native(comm, stream, cmds_packed, gathered, out):
    """gather_draw_list_packed with the collective inside the library (mcull_allgather_commands): ncclAllGather on the
    caller's communicator and stream + the rebase launch, two enqueues from C++ with no tensor bookkeeping in between.
    comm: RcclComm; stream: a torch.cuda.Stream; gathered [world * rows, 2] int32 and out [2, world] int64 are the
    buffers a frame loop keeps.  Returns (totals_all, cmds_all) like gather_draw_list_packed."""
    rows = cmds_packed.shape[0]
    with torch.cuda.device(cmds_packed.device):
        _pkg().allgather_commands(comm, stream.cuda_stream, cmds_packed.data_ptr(), gathered.data_ptr(), comm.world, rows - 1,
                                    out[0].data_ptr(), out[1].data_ptr())
    return out[0], gathered.view(comm.world, rows, 2)[:, : rows - 1, :]


def merge_depth(depth, group=None):
    """In place: the element-wise maximum over all ranks of `depth` (float32 tensor, the rank's own
    rasterised image, 0 where it drew nothing).  Depth is reverse-Z and never negative or NaN, so the maximum
    is the nearest surface of any rank - the merge the rasteriser's atomicMax performs inside a rank - and the
    result does not depend on the order of the reduction: it equals the unsharded image bit for bit."""
    dist.all_reduce(depth, op=dist.ReduceOp.MAX, group=group)
    return depth


class _DeviceArray:
    """A device buffer owned by the library, described for torch.as_tensor (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def sharded_frame(ctx, stream, globals_, width, height, view=0, prepass_tests=3, group=None, **cull_kwargs):
    """The culling work of one recorded frame (Engine.cpp:2381-2551) on a rank that holds one shard of the
    instances: cull with `prepass_tests`, rasterise the survivors, merge the depth images of all ranks
    (one all-reduce of width*height floats), build the pyramid of the whole scene in slot `view`, cull with
    all three tests (cull_kwargs are passed on to ctx.cull).  `stream` is the torch.cuda.Stream the context
    was created on: the collective is issued on it, between the rasteriser and the pyramid.  Returns the
    result of the second cull; gather_draw_list / gather_draw_list_packed then describe the global list."""
    ctx.cull(globals_, tests=prepass_tests, hiz_view=view)
    ctx.depth_rasterize(globals_, width, height)
    ptr, w, h = ctx.depth_buffer()
    depth = torch.as_tensor(_DeviceArray(ptr, (h, w), "<f4"), device="cuda:%d" % ctx.device)
    with torch.cuda.stream(stream):
        merge_depth(depth, group=group)
    ctx.hiz_build(ptr, view=view, width=width, height=height, mem=1)
    return ctx.cull(globals_, hiz_view=view, **cull_kwargs)
