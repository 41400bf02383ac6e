"""World-size-2 gloo tests (CPU): the multi-GPU path is query sharding + ONE broadcast of the index
image.  No compute kernels run here; what is checked is that every rank ends up with the identical
image and a usable handle, and that the shards tile the batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import genie_smem_amd as g
    from genie_smem_amd import parallel, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image = None
        if rank == 0:
            ref = synth.synth_ref(5000, 5)
            ix = g.GenieIndex.build(ref, 8)
            image = ix.serialize()
        buf = parallel.broadcast_image(image, src=0, device="cpu")
        hdr = parallel.header_of(buf)
        # every rank can open its copy (handle creation only: no device, so no launches here)
        ix2 = g.GenieIndex.from_image(buf)
        info = ix2.info()
        lo, hi = parallel.shard_bounds(1001, rank, world)
        digest = int(torch.sum(buf.to(torch.int64) * (torch.arange(buf.numel()) % 251 + 1)).item())
        q.put((rank, buf.numel(), digest, info["n"], info["K"], info["has_host"], lo, hi, bytes(hdr[:8].numpy())))
        with pytest.raises(RuntimeError):
            ix2.sa_interval(np.zeros((1, 4), np.uint8))          # no GPU -> loud failure, never a CPU path
        dist.barrier()
    finally:
        dist.destroy_process_group()                             # (no barrier here: a rank that failed must not hold the other)


def test_broadcast_and_shards_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = sorted(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:                                          # a rank that died must not leave its peer waiting
            if p.is_alive():
                p.terminate()
    (r0, n0, d0, nn0, k0, host0, lo0, hi0, m0), (r1, n1, d1, nn1, k1, host1, lo1, hi1, m1) = out
    assert (n0, d0, nn0, k0, m0) == (n1, d1, nn1, k1, m1) and nn0 == 5000 and k0 == 8
    assert m0 == (0x58444947454E4547).to_bytes(8, "little")
    assert host0 == 0 and host1 == 0                     # opened from the image: device-view handles
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)


def test_shard_bounds_tile_the_batch():
    import genie_smem_amd  # noqa: F401
    from genie_smem_amd import parallel
    for n in (0, 1, 7, 1000, 1001, 10 ** 6 + 3):
        for w in (1, 2, 4, 8):
            cuts = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(hi - lo for lo, hi in cuts) <= (n + w - 1) // w


def test_device_read_generator_on_cpu():
    """synth.reads_from_ref_device (what bench.py and the config-4 test generate their batches with), run on the CPU
    device: seeded, every read made of reference segments like the host generators' reads."""
    import torch
    from genie_smem_amd import synth
    ref = synth.synth_ref(20_000, 7)
    a = synth.reads_from_ref_device(ref, 3000, 150, 1005, device="cpu", chunk=1024)
    b = synth.reads_from_ref_device(torch.as_tensor(ref), 3000, 150, 1005, device="cpu", chunk=1024)
    assert a.dtype == torch.uint8 and tuple(a.shape) == (3000, 150) and torch.equal(a, b)
    assert not torch.equal(a, synth.reads_from_ref_device(ref, 3000, 150, 1006, device="cpu"))
    kmers = {ref[i:i + 12].tobytes() for i in range(len(ref) - 11)}

    def share(reads):                       # 12-mers of the reads that occur in the reference
        reads = np.asarray(reads)
        return np.mean([reads[r, i:i + 12].tobytes() in kmers for r in range(200) for i in range(0, 139, 3)])
    want = share(synth.reads_from_ref(ref, 200, 150, 5))
    assert abs(share(a.numpy()) - want) < 0.05 and want > 0.3
    assert share(synth.reads_random(200, 150, 5)) < 0.02
