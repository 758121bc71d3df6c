"""Multi-GPU support for the conv path: batch sharding + one-time weight broadcast.

The path shards over the batch dimension only: images are independent, so each rank (one process
per GPU) runs its own contiguous slice of the batch -- or, for batch-1 workloads, its own replica --
with no collective on the data path.  The single exchange is at setup: rank 0 packs the weights
(the plan's constant block: packed kernel + zero-point fold + scale/bias tables) and every other
rank receives the bytes with an RCCL broadcast over xGMI.  Blocks are coalesced into a few large
buckets so that the broadcast is a handful of big messages rather than one small one per layer
(xGMI links are point-to-point: message count, not bytes, is what costs at this size).

Everything here is written against a tiny "memory" interface so that the same code runs under
gloo on CPU in the tests (world_size 2) and under nccl(=RCCL) on MI355X.
"""
import zlib

import numpy as np

BUCKET_BYTES = 64 << 20


def shard_batch(total, world, rank):
    """Contiguous slice [start, stop) of a batch of `total` images owned by `rank`.
    The first total % world ranks take one extra image (ragged batches are allowed)."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def plan_buckets(sizes, bucket_bytes=BUCKET_BYTES):
    """Greedy coalescing of block sizes into buckets: list of lists of block indices."""
    buckets, cur, cur_bytes = [], [], 0
    for i, n in enumerate(sizes):
        if cur and cur_bytes + n > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(i)
        cur_bytes += n
    if cur:
        buckets.append(cur)
    return buckets


def broadcast_blocks(sizes, read_block, write_block, new_buffer, dist, src=0, bucket_bytes=BUCKET_BYTES):
    """Broadcast opaque byte blocks from rank `src` to every rank.

    sizes[i]               byte size of block i (identical on every rank)
    read_block(i, view)    copy block i into the torch uint8 tensor `view`       (used on src)
    write_block(i, view)   copy the torch uint8 tensor `view` into block i       (used elsewhere)
    new_buffer(nbytes)     uint8 tensor on the device the process group works on
    Returns the number of collectives issued.
    """
    rank = dist.get_rank()
    issued = 0
    for bucket in plan_buckets(sizes, bucket_bytes):
        total = sum(sizes[i] for i in bucket)
        buf = new_buffer(total)
        off = 0
        if rank == src:
            for i in bucket:
                read_block(i, buf[off:off + sizes[i]])
                off += sizes[i]
        dist.broadcast(buf, src=src)
        issued += 1
        if rank != src:
            off = 0
            for i in bucket:
                write_block(i, buf[off:off + sizes[i]])
                off += sizes[i]
    return issued


def _control_device(dist):
    """where the small control tensors of a collective live: the GPU for an nccl group, the host for gloo"""
    try:
        return "cuda" if dist.get_backend() == "nccl" else "cpu"
    except Exception:
        return "cpu"


def broadcast_plan_blocks(chain, torch, dist, hip, src=0):
    """RCCL broadcast of every layer's constant block of `chain` (workloads.LayerChain)."""
    blocks = chain.const_blocks()
    sizes = [n for _, n in blocks]
    dev = torch.device("cuda", torch.cuda.current_device())

    def read_block(i, view):
        hip.shl_mi355x_copy(view.data_ptr(), blocks[i][0], sizes[i], None)

    def write_block(i, view):
        hip.shl_mi355x_copy(blocks[i][0], view.data_ptr(), sizes[i], None)

    def new_buffer(n):
        return torch.empty(n, dtype=torch.uint8, device=dev)

    def synced_read(i, view):
        read_block(i, view)
        hip.shl_mi355x_stream_sync(None)

    n = broadcast_blocks(sizes, synced_read, write_block, new_buffer, dist, src)
    hip.shl_mi355x_stream_sync(None)
    torch.cuda.synchronize()
    return n


def device_bus_id(hip):
    """PCI bus id of this process's current device through the C-ABI ("" when there is none)"""
    import ctypes as C
    buf = C.create_string_buffer(32)
    return buf.value.decode() if hip.shl_mi355x_device_bus_id(buf, 32) == 0 else ""


def gather_bus_ids(torch, dist, mine):
    """every rank's PCI bus id, in rank order (32 bytes each over the bootstrap group)"""
    ctl = _control_device(dist)
    raw = mine.encode()[:32].ljust(32, b"\0")
    t = torch.tensor(list(raw), dtype=torch.uint8, device=ctl)
    out = [torch.zeros(32, dtype=torch.uint8, device=ctl) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [bytes(o.cpu().tolist()).rstrip(b"\0").decode() for o in out]


def shared_devices(bus_ids):
    """{bus id: [ranks]} for every device that more than one rank sits on ("" = a rank without a device)"""
    seen = {}
    for r, b in enumerate(bus_ids):
        seen.setdefault(b, []).append(r)
    return {b: rs for b, rs in seen.items() if len(rs) > 1 or b == ""}


LAST_BROADCAST = {}  # facts about the last broadcast_weights call of this process (bench.py puts them into its JSON line)
# bench.py installs a callable here: called (from a timer thread) with a message when the RCCL section of
# broadcast_weights -- ncclCommInitRank + the grouped ncclBroadcast, the one place a multi-GPU run can wait for a peer
# for ever -- has not returned after RCCL_STALL_SECONDS; the handler reports and ends the process
STALL_HANDLER = None
RCCL_STALL_SECONDS = 240.0


class _StallWatch:
    def __init__(self, what):
        import threading
        self.t = None
        if STALL_HANDLER is not None:
            self.t = threading.Timer(RCCL_STALL_SECONDS, STALL_HANDLER, args=("%s did not return within %.0f s" % (what, RCCL_STALL_SECONDS),))
            self.t.daemon = True

    def __enter__(self):
        if self.t:
            self.t.start()
        return self

    def __exit__(self, *exc):
        if self.t:
            self.t.cancel()
        return False


def broadcast_weights(chain, torch, dist, hip, opt, rank, world, src=0, prefer_c=True, bus_id=None):
    """The one-time weight broadcast of `chain`.  Preferred: RCCL behind the C boundary -- rank `src` draws an
    ncclUniqueId through the C-ABI, the 128 bytes travel over the process group that already exists (bootstrap
    only), every rank creates the communicator and calls shl_mi355x_bcast_const_blocks (host code stays C, as
    BASELINE north_star asks).  All ranks agree first on whether that path is usable; otherwise (no librccl,
    or a gloo group on one device in the tests) the same bytes go through torch.distributed.  Returns a
    description of the path taken."""
    import ctypes as C
    why = "disabled"
    ctl = _control_device(dist)  # the process group is bootstrap only: small CPU tensors under gloo
    # one device per rank is what ncclCommInitRank needs; ranks that share a device (or have none) must find out HERE,
    # over the bootstrap group, not inside a collective that then waits for a peer that can never join
    bus_ids = gather_bus_ids(torch, dist, device_bus_id(hip) if bus_id is None else bus_id)
    shared = shared_devices(bus_ids)
    LAST_BROADCAST.clear()
    LAST_BROADCAST.update(bus_ids=bus_ids, distinct_devices=len(set(b for b in bus_ids if b)), rccl_nranks=None)
    if shared:
        prefer_c = False
        why = "RCCL refused before ncclCommInitRank: " + "; ".join(
            ("ranks %s have no device" % rs) if b == "" else ("ranks %s share device %s" % (rs, b)) for b, rs in sorted(shared.items()))
    if prefer_c:
        ok = torch.tensor([int(hip.shl_mi355x_comm_available())], dtype=torch.int32, device=ctl)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        why = "librccl not found on every rank"
        if int(ok.item()) == 1:
            uid = (C.c_ubyte * 128)()
            if rank == src and hip.shl_mi355x_comm_unique_id(uid) != 0:
                raise RuntimeError("ncclGetUniqueId failed: " + hip.shl_mi355x_last_error().decode())
            t = torch.tensor(list(uid), dtype=torch.uint8, device=ctl)
            dist.broadcast(t, src=src)
            uid = (C.c_ubyte * 128)(*t.cpu().tolist())
            comm = C.c_void_p()
            with _StallWatch("ncclCommInitRank (rank %d of %d)" % (rank, world)):
                rc = hip.shl_mi355x_comm_create(uid, rank, world, C.byref(comm))
                err = "" if rc == 0 else hip.shl_mi355x_last_error().decode()
                good = torch.tensor([int(rc == 0)], dtype=torch.int32, device=ctl)
                dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if int(good.item()) == 1:
                nr, me, dv = C.c_int32(), C.c_int32(), C.c_int32()
                if hip.shl_mi355x_comm_info(comm, C.byref(nr), C.byref(me), C.byref(dv)) == 0:
                    LAST_BROADCAST.update(rccl_nranks=nr.value, rccl_rank=me.value, rccl_device=dv.value)
                n = len(chain.entries)
                params = (C.c_void_p * n)(*[C.cast(e["params"], C.c_void_p) for e in chain.entries])
                with _StallWatch("the grouped ncclBroadcast of %d blocks (rank %d of %d)" % (n, rank, world)):
                    rc = opt.shl_mi355x_bcast_const_blocks(comm, params, n, src, chain.sess)
                hip.shl_mi355x_comm_destroy(comm)
                if rc != 1:
                    raise RuntimeError("shl_mi355x_bcast_const_blocks failed: " + hip.shl_mi355x_last_error().decode())
                LAST_BROADCAST.update(transport="RCCL ncclBroadcast")
                return "RCCL ncclBroadcast behind the C-ABI (shl_mi355x_bcast_const_blocks), %d blocks in one group, communicator of ranks 0..%d" % (n, world - 1)
            if comm:
                hip.shl_mi355x_comm_destroy(comm)
            why = "ncclCommInitRank failed on some rank" + (" (this one: %s)" % err if err else "")
            LAST_BROADCAST.update(rccl_error=why)
    n = broadcast_plan_blocks(chain, torch, dist, hip, src=src)
    # tables and the code path chosen for them come from the same rank: adopt the root's flags records
    cnt = len(chain.entries)
    params = (C.c_void_p * cnt)(*[C.cast(e["params"], C.c_void_p) for e in chain.entries])
    if opt.shl_mi355x_params_adopt_blocks(params, cnt, chain.sess) != 1:
        raise RuntimeError("shl_mi355x_params_adopt_blocks failed: " + hip.shl_mi355x_last_error().decode())
    LAST_BROADCAST.update(transport="torch.distributed %s fallback" % dist.get_backend())
    if prefer_c and "rccl_error" not in LAST_BROADCAST:
        LAST_BROADCAST.update(rccl_error=why)
    return "torch.distributed broadcast, %d buckets (%s)" % (n, why)


def broadcast_weights_one_rank(chain, hip, opt):
    """All of the C / RCCL path that ONE process can exercise: ncclGetUniqueId -> ncclCommInitRank(world 1) ->
    shl_mi355x_bcast_const_blocks over the chain's plans -> adoption of the flags records -> destroy.  Used by
    `bench.py --total-batch B` on a single GPU, so that the sharded configuration runs through the same entry points a
    multi-GPU launch takes."""
    import ctypes as C
    if hip.shl_mi355x_comm_available() != 1:
        return "no librccl on this box: " + hip.shl_mi355x_last_error().decode()
    uid = (C.c_ubyte * 128)()
    if hip.shl_mi355x_comm_unique_id(uid) != 0:
        raise RuntimeError("ncclGetUniqueId failed: " + hip.shl_mi355x_last_error().decode())
    comm = C.c_void_p()
    if hip.shl_mi355x_comm_create(uid, 0, 1, C.byref(comm)) != 0:
        raise RuntimeError("ncclCommInitRank failed: " + hip.shl_mi355x_last_error().decode())
    n = len(chain.entries)
    params = (C.c_void_p * n)(*[C.cast(e["params"], C.c_void_p) for e in chain.entries])
    rc = opt.shl_mi355x_bcast_const_blocks(comm, params, n, 0, chain.sess)
    hip.shl_mi355x_comm_destroy(comm)
    if rc != 1:
        raise RuntimeError("shl_mi355x_bcast_const_blocks failed: " + hip.shl_mi355x_last_error().decode())
    return "RCCL ncclBroadcast behind the C-ABI (shl_mi355x_bcast_const_blocks), %d blocks in one group, communicator of rank 0 alone" % n


def checksum_bytes(arr):
    return zlib.crc32(np.ascontiguousarray(arr).view(np.uint8).tobytes())


def assert_replicas_agree(chain, torch, dist, hip):
    """After the weight broadcast every rank must hold identical plans: run the chain on a
    rank-independent probe image and compare output checksums across ranks."""
    sums = []
    for e in chain.entries:
        n = int(np.prod(e["in_dims"])) * chain.esize
        probe = (np.arange(n, dtype=np.int64) * 37 % 251 - 125).astype(np.int8)
        if chain.dtype != "int8":
            probe = probe[: n // 2].astype(np.float16) / 64
        hip.shl_mi355x_upload(e["d_in"], probe.ctypes.data, probe.nbytes, None)
        hip.shl_mi355x_stream_sync(None)
        chain.opt.shl_mi355x_set_stream(None)
        rc = e["run"](*e["args"])
        assert rc == 1
        out = np.empty(int(np.prod(e["out_dims"])) * chain.esize, dtype=np.uint8)
        hip.shl_mi355x_stream_sync(None)
        hip.shl_mi355x_download(out.ctypes.data, e["d_out"], out.nbytes, None)
        hip.shl_mi355x_stream_sync(None)
        sums.append(checksum_bytes(out))
    mine = torch.tensor(sums, dtype=torch.int64, device=_control_device(dist))
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not bool((lo == hi).all()):
        raise RuntimeError("replicas disagree after the weight broadcast: %s" % (lo != hi).nonzero().flatten().tolist())
    return True
