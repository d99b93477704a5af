"""Disparity-sharded census (+ SGM) over the GPUs of one node: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI) for the single exchange of the pipeline.

    rank r:  keys_r = svh_census_shard_keys(shard r of the disparity range)      (H, W, 2) int32, 8 B / pixel
             all_reduce(keys, MIN)                                              RCCL; 4 B / pixel when the second plane is global
             disp   = svh_census_shard_finish(keys)                             replicated, bit-identical to 1 GPU

The second key of a pixel belongs to the disparities that pay Pout (sgm.h:287-289).  When all of them look outside the target
image (RightToLeft, source width + first offset >= target width: the benchmark's geometry) their costs are equal on every
shard, the last index of the whole range wins, and svh_census_shard_keys writes exactly that: plane 1 needs no exchange and
`exchange_keys` reduces plane 0 only.

Why one int32 MIN all-reduce is enough: in the Cost branch as the reference computes it a pass couples
the disparities of a pixel only through min_d [c + (c [+Pout])]; with integer census costs that minimum and the winner
are both functions of the two regional minima (cost, last index) the keys carry.

Streams of frames (video, a survey's image pairs): ShardedStereoPipeline keeps one exchange in flight, so that the
all-reduce of frame k (RCCL's own stream, xGMI) runs under the key kernels of frame k + 1 on the compute stream.
"""
import torch.distributed as dist

from . import correlation as _c


def shard_range(total, rank, world, align=1):
    """Contiguous split of `total` disparities; the first ranks get one more unit when it does not divide.  `align` > 1 splits in
    units of that many disparities when total is a multiple of it and every rank still gets a unit (32 keeps every shard a whole
    number of the matrix-core sweep's row tiles); otherwise single disparities."""
    total, world, align = int(total), int(world), int(align)
    if align > 1 and total % align == 0 and total // align >= world:
        begin, count = shard_range(total // align, rank, world)
        return begin * align, count * align
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


class _KeyExchange:
    """MIN all-reduce of the key planes that need it, in place in `keys` once wait() returns."""

    def __init__(self, keys, plane0_only, group, async_op):
        self.keys, self.plane0 = keys, None
        if plane0_only:
            self.plane0 = keys[..., 0].contiguous()  # collectives want dense buffers: 4 B / pixel travel instead of 8
            self.work = dist.all_reduce(self.plane0, op=dist.ReduceOp.MIN, group=group, async_op=async_op)
        else:
            self.work = dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group, async_op=async_op)

    def wait(self):
        if self.work is not None:
            self.work.wait()  # RCCL: the compute stream waits for the exchange; gloo: the host does
        if self.plane0 is not None:
            self.keys[..., 0].copy_(self.plane0)
        return self.keys


def exchange_keys(keys, plane0_only, group=None, async_op=False):
    """The one exchange of the protocol; returns an object whose wait() gives the reduced keys."""
    return _KeyExchange(keys, plane0_only, group, async_op)


# ---- the exchange through the C ABI (svh_census_exchange_keys) ---------------------------------------------------------------------
# What a C++ host does (libstevi_amd/include/correlation/sharded.h, tools/bench_sharded.cpp) from Python: a communicator of this
# process's own, made with the RCCL instance PyTorch has already loaded, and the all-reduce enqueued by the library on the stream its
# kernels run on.  torch.distributed only carries the 128-byte unique id.  Used by tests and by `bench.py --c-abi-exchange`; the
# pipelines above keep torch.distributed's own collective (its communicator, its stream, its overlap).
class RcclCommunicator:
    """ncclComm_t for (rank, world) created through ctypes; `handle` is what svh_census_exchange_keys takes."""

    def __init__(self, rank=None, world=None, group=None, device=None):
        import ctypes as C
        import os
        import torch
        self._C = C
        self.rank = (dist.get_rank(group) if dist.is_initialized() else 0) if rank is None else int(rank)
        self.world = (dist.get_world_size(group) if dist.is_initialized() else 1) if world is None else int(world)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._lib = C.CDLL(path if os.path.exists(path) else "librccl.so.1")  # (the instance already in the process, by soname)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UniqueId()
        self._lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self._lib.ncclGetErrorString.restype = C.c_char_p
        if self.rank == 0:
            self._ok(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        if device is not None:
            torch.cuda.set_device(device)
        comm = C.c_void_p()
        self._ok(self._lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.handle = comm.value

    def _ok(self, status, what):
        if status != 0:
            raise RuntimeError(f"{what}: {self._lib.ncclGetErrorString(status).decode()}")

    def destroy(self):
        if getattr(self, "handle", None):
            self._lib.ncclCommDestroy(self._C.c_void_p(self.handle))
            self.handle = None


def exchange_keys_rccl(keys, plane0_only, comm):
    """svh_census_exchange_keys: the int32 MIN all-reduce of `keys` ((H, W, 2) int32 on the GPU) in place over `comm`
    (RcclCommunicator), enqueued on the stream of the tensor's context; returns `keys`."""
    import ctypes as C
    from . import _capi
    ctx = _c.context_for(keys)
    d = _c._desc(keys)
    _c._check(ctx, _capi.load().svh_census_exchange_keys(ctx, C.c_void_p(comm.handle), C.byref(d), int(bool(plane0_only))))
    return keys


def stereoMatchSharded(img_l, img_r, h_radius, v_radius, disp_width, group=None, **kw):
    """Census + SGM with the disparity range split over the ranks of `group`.  Every rank passes the same images
    (resident on its own GPU) and gets the same disparity map back.  kw: dDir, sgmDirections, P1, P2, Pout, margins,
    refineKernel, refine_h_radius, refine_v_radius."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    _, D = _c._search_range(disp_width)
    shard = shard_range(D, rank, world)
    if shard[1] == 0:
        raise ValueError("more ranks than disparities")
    keys_kw = {k: v for k, v in kw.items() if k in ("dDir", "sgmDirections", "P1", "P2", "Pout", "margins", "matchFunc")}
    keys = _c.censusShardKeys(img_l, img_r, h_radius, v_radius, disp_width, shard, **keys_kw)
    if keys.shape[0] == 0:  # the reference's empty result (row / channel mismatch): the same on every rank, nothing to exchange
        return {"disp": _c._empty_like(keys, 2, "i32")}
    if world > 1:
        plane0_only = _c.censusShardRegion1IsGlobal(img_l, img_r, disp_width, kw.get("dDir", _c.dispDirection.RightToLeft))
        keys = exchange_keys(keys, plane0_only, group).wait()
    return _c.censusShardFinish(img_l, img_r, keys, h_radius, v_radius, disp_width, **kw)


class ShardedStereoPipeline:
    """stereoMatchSharded over a stream of frames with the exchange of one frame overlapped with the compute of the
    next:

        submit(frame k):  keys_k = censusShardKeys(...)                 compute stream
                          work_k = all_reduce(keys_k, MIN, async_op)    RCCL stream, starts when keys_k are written
                          return finish(frame k - 1)                    waits for work_{k-1} only

    Results are the ones stereoMatchSharded returns, one submit late; flush() returns the last one.  The keys of a frame
    are a fresh tensor per submit, so the frame in flight is never overwritten."""

    def __init__(self, h_radius, v_radius, disp_width, group=None, align=1, **kw):
        self.h_radius, self.v_radius, self.disp_width, self.group, self.kw = h_radius, v_radius, disp_width, group, kw
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        _, D = _c._search_range(disp_width)
        self.shard = shard_range(D, self.rank, self.world, align)
        if self.shard[1] == 0:
            raise ValueError("more ranks than disparities")
        self._in_flight = None

    # the two device stages (overridable: the CPU tests drive the pipeline with the numpy restatement of the protocol)
    def _keys(self, img_l, img_r):
        keys_kw = {k: v for k, v in self.kw.items() if k in ("dDir", "sgmDirections", "P1", "P2", "Pout", "margins", "matchFunc")}
        return _c.censusShardKeys(img_l, img_r, self.h_radius, self.v_radius, self.disp_width, self.shard, **keys_kw)

    def _finish(self, img_l, img_r, keys):
        return _c.censusShardFinish(img_l, img_r, keys, self.h_radius, self.v_radius, self.disp_width, **self.kw)

    def _plane0_only(self, img_l, img_r):
        return _c.censusShardRegion1IsGlobal(img_l, img_r, self.disp_width, self.kw.get("dDir", _c.dispDirection.RightToLeft))

    def _complete(self, frame):
        img_l, img_r, keys, exchange = frame
        if keys.shape[0] == 0:
            return {"disp": _c._empty_like(keys, 2, "i32")}
        if exchange is not None:
            keys = exchange.wait()
        return self._finish(img_l, img_r, keys)

    def submit(self, img_l, img_r):
        """Start frame k; returns the result of frame k - 1 (None for the first frame)."""
        keys = self._keys(img_l, img_r)
        if keys.shape[0] == 0:  # the reference's empty result: nothing to exchange, nothing to finish
            previous, self._in_flight = self._in_flight, None
            done = self._complete(previous) if previous is not None else None
            self._in_flight = (img_l, img_r, keys, None)
            return done
        exchange = exchange_keys(keys, self._plane0_only(img_l, img_r), self.group, async_op=True) if self.world > 1 else None
        previous, self._in_flight = self._in_flight, (img_l, img_r, keys, exchange)
        return self._complete(previous) if previous is not None else None

    def flush(self):
        """Result of the last submitted frame (None when nothing is in flight)."""
        previous, self._in_flight = self._in_flight, None
        return self._complete(previous) if previous is not None else None


# ---- row bands -------------------------------------------------------------------------------------------------------
# The same disparity map split by ROWS instead of disparities.  In the integer-exact regime the winner of a pixel depends on the
# pixel's own costs and on its position only (svh_census_band_match, include/stevi_hip.h), so the bands are independent: no exchange
# is needed to compute them, and rank r simply holds rows band_range(H, r, world) of the map.  gather=True replicates the map on
# every rank with one all_gather (4 B / pixel over all ranks, half of the key all-reduce), kept in flight under the next frame.
def band_range(rows, rank, world):
    """Contiguous split of the image rows; the first ranks get one more row when it does not divide."""
    return shard_range(rows, rank, world)


class RowBandStereoPipeline:
    """submit(img_l, img_r) -> this rank's band of the disparity map, (rows_r, W) int32 [gather=True: the whole map of the PREVIOUS
    frame, None for the first; flush() returns the last one].  compute=None runs svh_census_band_match; a callable
    (img_l, img_r, (begin, count)) -> band replaces it (the CPU tests pass the oracle)."""

    def __init__(self, h_radius, v_radius, disp_width, group=None, gather=False, compute=None, **kw):
        self.args = (h_radius, v_radius, disp_width)
        self.kw = {k: v for k, v in kw.items() if k in ("dDir", "sgmDirections", "P1", "P2", "Pout", "margins", "matchFunc")}
        self.group, self.gather, self.compute = group, gather, compute
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.pending = None

    def rows_of(self, img_l, img_r, rank=None):
        dDir = self.kw.get("dDir", _c.dispDirection.RightToLeft)
        src = img_r if int(dDir) == _c.dispDirection.RightToLeft else img_l
        return band_range(src.shape[0], self.rank if rank is None else rank, self.world)

    def _band(self, img_l, img_r):
        rows = self.rows_of(img_l, img_r)
        if self.compute is not None:
            return self.compute(img_l, img_r, rows)
        return _c.censusBandMatch(img_l, img_r, *self.args, rows, **self.kw)

    def _collect(self):
        if self.pending is None:
            return None
        work, out, counts = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        import torch
        return torch.cat([out[r, :c] for r, c in enumerate(counts)], 0)

    def submit(self, img_l, img_r):
        band = self._band(img_l, img_r)
        if not self.gather or self.world == 1:
            return band
        import torch
        done = self._collect()
        counts = [self.rows_of(img_l, img_r, r)[1] for r in range(self.world)]
        padded = band
        if band.shape[0] != max(counts):  # equal contributions (bands differ by at most one row)
            padded = torch.zeros((max(counts), band.shape[1]), dtype=band.dtype, device=band.device)
            padded[:band.shape[0]] = band
        out = torch.empty((self.world, max(counts), band.shape[1]), dtype=band.dtype, device=band.device)
        work = dist.all_gather([out[r] for r in range(self.world)], padded.contiguous(), group=self.group, async_op=True)  # (equal sizes: gloo too)
        self.pending = (work, out, counts)
        return done

    def flush(self):
        return self._collect()
