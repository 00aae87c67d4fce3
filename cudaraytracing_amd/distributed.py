"""Multi-GPU sharding of the frame: pixel tiles interleaved over ranks, one
RCCL all-gather of the compact per-rank tile buffers, then a de-interleave.

The reference has no multi-GPU path (config_CUDA hard-selects device 0,
src/main.cu:99-100); pixels are independent and the RNG is keyed by the global
pixel index, so the image does not depend on the partition.

Tile t (8x8 pixels, row-major over ceil(W/8) x ceil(H/8)) belongs to rank
t % world; rank r stores its k-th tile (t = k*world + r) at slots [64k, 64k+64).
"""
import numpy as np

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def local_tiles(width, height, world):
    tx, ty = tile_grid(width, height)
    return (tx * ty + world - 1) // world


def untile_numpy(gathered, width, height):
    """gathered: (world, slots, C) array of per-rank compact tile buffers -> (H, W, C) image."""
    world, slots, ch = gathered.shape
    tx, ty = tile_grid(width, height)
    lt = slots // 64
    a = gathered.reshape(world, lt, TILE, TILE, ch)
    a = np.transpose(a, (1, 0, 2, 3, 4)).reshape(lt * world, TILE, TILE, ch)[:tx * ty]
    a = a.reshape(ty, tx, TILE, TILE, ch)
    a = np.transpose(a, (0, 2, 1, 3, 4)).reshape(ty * TILE, tx * TILE, ch)
    return np.ascontiguousarray(a[:height, :width])


def untile_torch(gathered, width, height):
    """Same as untile_numpy for a torch tensor (stays on its device)."""
    world, slots, ch = gathered.shape
    tx, ty = tile_grid(width, height)
    lt = slots // 64
    a = gathered.reshape(world, lt, TILE, TILE, ch).permute(1, 0, 2, 3, 4).reshape(lt * world, TILE, TILE, ch)[:tx * ty]
    a = a.reshape(ty, tx, TILE, TILE, ch).permute(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, ch)
    return a[:height, :width].contiguous()


_hip_runtime_checked = False


def _check_single_hip_runtime():
    """torch bundles its own libamdhip64.so.7; libcrt.so links the system one.  The loader
    shares one copy only if torch is imported BEFORE libcrt.so is loaded (it matches the
    request against the SONAME of what is already mapped).  Two HIP runtimes in one process
    cannot see each other's allocations, so refuse to continue."""
    global _hip_runtime_checked
    if _hip_runtime_checked:  # (once per process: libcrt.so and torch are both loaded by the time of the first frame)
        return
    paths = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                paths.add(line.split()[-1])
    _hip_runtime_checked = len(paths) <= 1
    if len(paths) > 1:
        raise RuntimeError("two HIP runtimes are mapped (%s): import torch before using cudaraytracing_amd"
                           % ", ".join(sorted(paths)))


def gather_image(local, width, height, world, group=None):
    """All-gathers the ranks' compact tile buffers (torch tensors of shape (slots, C)) and returns the
    de-interleaved (H, W, C) image on every rank.  One collective: the message is tiny (180 KB per rank
    at 800x600), so it is latency bound; a single all-gather lets every xGMI link carry each peer's
    slice once."""
    import torch
    import torch.distributed as dist
    if world > 1:
        if dist.get_backend(group) == "gloo" and local.is_cuda:  # test configuration only: gloo gathers host tensors
            host = torch.empty((world,) + tuple(local.shape), dtype=local.dtype)
            dist.all_gather_into_tensor(host.view(-1), local.cpu().reshape(-1), group=group)
            gathered = host.to(local.device)
        else:
            gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(gathered.view(-1), local.reshape(-1), group=group)
    else:
        gathered = local.unsqueeze(0)
    return untile_torch(gathered, width, height)


def render_sharded(render, eye, inv_view, fov_y, width, height, rank, world, device, group=None, want_stats=True):
    """One process per GPU: renders this rank's tiles into a torch uint8 tensor, all-gathers
    the compact buffers over RCCL (torch.distributed backend "nccl") and returns
    (image tensor (H, W, 3) on `device`, stats of this rank).  world == 1 skips the collective."""
    import torch
    import torch.distributed as dist
    from .api import shard_slots

    _check_single_hip_runtime()

    slots = shard_slots(width, height, rank, world)
    local = torch.empty((slots, 3), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None
    stats = render.run_view_device(eye, inv_view, fov_y, local.data_ptr(), None, stream, rank=rank, world=world,
                                   tiled=True, want_stats=want_stats, width=width, height=height)
    return gather_image(local, width, height, world, group), stats


class FramePipeline:
    """Two frames in flight per rank.  Every launch of the persistent render kernel ends with its waves running their private
    path pools dry (about 2 ms): with one frame at a time that tail is idle GPU, 1.7 % of a C2 frame on
    one GPU and 12 % of a rank's share on eight.  The pipeline alternates two device replicas of the scene (two crt_scene handles:
    path pools, per-path radiance and accumulators are per handle) on two HIP streams, so the workgroups of frame i+1 are
    dispatched into the compute units as the waves of frame i retire; the tile gather of a frame is enqueued on the frame's
    own stream behind its kernels.  Frames complete in submission order; nothing synchronizes the host until `drain()` or until
    a handle is reused (by then its previous frame has long finished).

    submit() returns nothing; finished frames are collected in `self.done` as (image tensor, kernel ms) in submission order."""

    def __init__(self, renders, eye, inv_view, fov_y, width, height, rank, world, device, group=None, depth=2):
        import torch
        self.renders = list(renders)[:max(1, depth)]
        self.args = (eye, inv_view, fov_y, width, height, rank, world, device, group)
        # HIP multiplexes streams onto a few hardware queues and kernels of one queue run back to back: two streams of the SAME priority
        # landed on one queue here (rocprofv3 kernel trace: Queue_Id equal, no overlap).  Streams of different priorities live on
        # different queues, so the two frames in flight alternate between a normal and a high-priority stream.
        self.streams = [torch.cuda.Stream(device=device, priority=(0 if k % 2 == 0 else -1)) for k in range(len(self.renders))]
        self.pending = [None] * len(self.renders)   # per handle: image tensor of the frame in flight
        self.ready = [None] * len(self.renders)     # per handle: event recorded behind the frame's last operation (the de-interleave that follows the gather)
        self.next = 0
        self.done = []
        self.kernel_ms = []
        self.kernel_ms_missing = 0   # frames of the fallback pipeline (no megakernel launch to time)

    def _retire(self, h):
        if self.pending[h] is not None:
            # The all-gather runs on ProcessGroupNCCL's own stream; the synchronous call makes the frame's stream wait for it, and the
            # de-interleave that produces the image is enqueued on the frame's stream behind that wait.  The event recorded after it is
            # what publishing the image waits for.
            if self.ready[h] is not None:
                self.ready[h].synchronize()
            self.streams[h].synchronize()
            from ._capi import CrtError, ERR_INVALID_ARG
            try:
                ms, _ = self.renders[h].last_launch_ms()
            except CrtError as e:
                # the one expected failure: the wavefront fallback pipeline records no megakernel launch (CRT_ERR_INVALID_ARG,
                # "no frame has been rendered by the megakernel"); anything else -- a HIP error of an unsynchronised stream or a
                # faulted device -- is a real error and is raised
                if e.status != ERR_INVALID_ARG or "no frame has been rendered by the megakernel" not in str(e):
                    raise
                ms = None
                self.kernel_ms_missing += 1
            self.kernel_ms.append(ms)
            self.done.append(self.pending[h])
            self.pending[h] = None

    def submit(self):
        import torch
        from .api import shard_slots
        eye, inv_view, fov_y, width, height, rank, world, device, group = self.args
        h = self.next
        self.next = (self.next + 1) % len(self.renders)
        self._retire(h)   # (the frame submitted two submissions ago: finished unless the pipeline is starved)
        s = self.streams[h]
        with torch.cuda.stream(s):
            slots = shard_slots(width, height, rank, world)
            local = torch.empty((slots, 3), dtype=torch.uint8, device=device)
            self.renders[h].run_view_device(eye, inv_view, fov_y, local.data_ptr(), None, s.cuda_stream, rank=rank, world=world, tiled=True,
                                            want_stats=False, width=width, height=height)
            self.pending[h] = gather_image(local, width, height, world, group)
            ev = torch.cuda.Event()
            ev.record(s)
            self.ready[h] = ev

    def drain(self):
        for k in range(len(self.renders)):
            self._retire((self.next + k) % len(self.renders))
