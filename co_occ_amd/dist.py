"""Multi-GPU layer.  The path shards at sample level (one scene per GPU, the
reference's own ``samples_per_gpu=1`` data parallelism) and, for rendering, at ray level.  The
only data-path collective is one RCCL all-gather of the packed rendered maps per step; there is
none in the reference's eval forward, it is a requirement of the build's
``north_star``.  One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI),
"gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pin_rank_threads(local_rank, local_world):
    """Give each rank of a node its own slice of the host cores this process may use (its issuing thread + search helper threads
    then do not migrate onto the cores of the other ranks: 8 ranks x 2-4 threads on a CPU-capped node would otherwise be a
    stampede).  No-op when there are fewer cores than ranks or the platform has no affinity call.  Returns the cores kept."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except Exception:
        return None
    if local_world <= 1 or len(cores) < local_world:
        return cores
    per = len(cores) // local_world
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except Exception:
        return cores
    return mine


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_maps(rgbs, depths):
    """[N,H,W,3] + [N,H,W] -> one contiguous [N,H,W,4] buffer (a single collective per step)."""
    return torch.cat([rgbs, depths.unsqueeze(-1)], dim=-1).contiguous()


def unpack_maps(packed):
    return packed[..., :3], packed[..., 3]


def all_gather_maps(rgbs, depths):
    """Config 4: every rank rendered its own scene; gather all ranks' maps.
    Returns (rgbs [world,N,H,W,3], depths [world,N,H,W])."""
    packed = pack_maps(rgbs, depths)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rgbs.unsqueeze(0), depths.unsqueeze(0)
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(packed.shape), device=packed.device, dtype=packed.dtype)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed)
        out = torch.stack(parts)
    else:
        dist.all_gather_into_tensor(out, packed)
    return unpack_maps(out)


class PendingMaps:
    """An all-gather of rendered maps in flight (all_gather_maps_async): ``wait()`` -> (rgbs [world,N,H,W,3],
    depths [world,N,H,W])."""

    def __init__(self, work, out, parts, packed):
        self.work, self.out, self.parts, self.packed = work, out, parts, packed

    def host_wait(self, poll_s=50e-6):
        """Wait for the collective ON THE HOST (sleep-polling ``is_completed``): no stream is made to wait for RCCL's stream, so no
        queue sits on an unsatisfied cross-queue wait while the collective runs.  Then as ``wait()``."""
        if self.work is not None:
            import time
            try:
                while not self.work.is_completed():
                    time.sleep(poll_s)
            except (RuntimeError, NotImplementedError):      # a backend without completion queries: block the host in wait()
                self.work.wait()
                torch.cuda.current_stream().synchronize()
            self.work = None
        return self.wait()

    def wait(self):
        if self.work is not None:
            self.work.wait()             # the CURRENT stream waits for the collective; no host block with RCCL
            self.work = None
        if self.parts is not None:
            self.out = torch.stack(self.parts)
            self.parts = None
        return unpack_maps(self.out)


def all_gather_maps_async(rgbs, depths):
    """all_gather_maps without making the issuing stream wait: RCCL runs the collective on its own stream once the
    maps are ready, and the next sample's dense stage is enqueued behind the render kernels, not behind the
    collective ('overlappable with the next sample').  Call ``.wait()`` on the result before reading."""
    packed = pack_maps(rgbs, depths)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return PendingMaps(None, packed.unsqueeze(0), None, packed)
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(packed) for _ in range(world)]
        return PendingMaps(dist.all_gather(parts, packed, async_op=True), None, parts, packed)
    out = torch.empty((world,) + tuple(packed.shape), device=packed.device, dtype=packed.dtype)
    return PendingMaps(dist.all_gather_into_tensor(out, packed, async_op=True), out, None, packed)


class GatherThread:
    """The rank's map all-gathers, issued by ONE host thread that first waits -- on the host, sleeping -- for the completion
    event of the replay that rendered the maps, then enqueues pack + ``all_gather_into_tensor`` on its own stream.

    Why not ``all_gather_maps_async`` right after the replay is issued: RCCL's stream would then sit on an unsatisfied wait for the
    replay's event for the whole dense stage (~10 ms), and a queue whose head is a pending cross-queue wait slows the dispatch of
    every other queue of the device (15-20 % of the serving loop's throughput, profiles/r6_serving_probe_events.txt).  One
    thread issues every collective of the rank in sample order, so all ranks issue them in the same order; at most one is in
    flight (the next one's stream waits for it), each overlaps the following samples' dense stages.  ``drain()`` before any
    collective issued by another thread (barriers, the timing all-reduce)."""

    def __init__(self, device):
        import queue
        import threading
        self.device, self.q, self.err, self.last = device, queue.Queue(), None, None
        self.t = threading.Thread(target=self._run, daemon=True, name="coocc-gather")
        self.t.start()

    def submit(self, out, event):
        self.q.put((out, event))

    def _run(self):
        torch.cuda.set_device(self.device)
        stream = torch.cuda.Stream(device=self.device)
        pending = None
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                if isinstance(item, tuple) and item[0] == "drain":
                    with torch.cuda.stream(stream):
                        if pending is not None:
                            self.last = pending.host_wait()
                            pending = None
                    stream.synchronize()
                    item[1].set()
                    continue
                out, event = item
                event.synchronize()                       # host-side: the maps exist (device-scope release done) from here on
                with torch.cuda.stream(stream):
                    if pending is not None:
                        self.last = pending.host_wait()   # at most one collective in flight; waited for on the host, not by a stream
                    pending = all_gather_maps_async(out["rgbs"], out["depths"])
            except Exception as e:                        # surfaced by drain()
                self.err = e
                if isinstance(item, tuple) and item and item[0] == "drain":
                    item[1].set()
            finally:
                self.q.task_done()

    def drain(self):
        """Every submitted gather has been issued and has completed.  Returns the last gathered (rgbs, depths)."""
        import threading
        done = threading.Event()
        self.q.put(("drain", done))
        done.wait()
        if self.err is not None:
            e, self.err = self.err, None
            raise e
        return self.last

    def close(self):
        self.q.put(None)
        self.t.join(timeout=10)


def gather_ray_shards(local_maps, n_rows_total):
    """Config 5 (ray-sharded render of ONE scene): each rank rendered a contiguous chunk of the
    flattened (camera,row) space; rows are padded to the largest chunk for the collective and
    trimmed afterwards.  local_maps [rows_local, W, 4] -> [n_rows_total, W, 4]."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_maps
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_rows_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local_maps.shape[1:]), device=local_maps.device, dtype=local_maps.dtype)
    pad[: local_maps.shape[0]] = local_maps
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    else:
        # RCCL: ONE all_gather_into_tensor over the equal-padded chunks (a list all_gather is world point-to-point copies)
        out = torch.empty((world,) + tuple(pad.shape), device=pad.device, dtype=pad.dtype)
        dist.all_gather_into_tensor(out, pad)
        parts = list(out.unbind(0))
    if all(hi - lo == mx for lo, hi in sizes):
        return torch.cat(parts, 0) if dist.get_backend() == "gloo" else out.reshape((world * mx,) + tuple(pad.shape[1:]))
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
