"""Multi-GPU path: root-tile-column shards combined by one SUM reduce.

CPU: the combine protocol at world size 2 over gloo, partial images cut from the oracle's frame
with the same ownership rule the device uses.  GPU: the device's own shards are disjoint, zero
elsewhere and sum to the single-GPU frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fidget_amd import dist as D  # noqa: E402

SIZE = 256
MODEL = os.path.join(ROOT, "models", "colonnade.vm")


def test_root_tile_rule():
    assert [D.root_tile(s) for s in (8, 9, 16, 33, 64, 65, 128, 129, 1024)] == [8, 16, 16, 64, 64, 128, 128, 128, 128]


def test_owner_map_partitions_the_image():
    for w, h, world in ((256, 256, 2), (1024, 1024, 8), (300, 200, 3)):
        own = D.owner_map(w, h, 128, world)
        assert own.shape == (h, w) and own.min() == 0 and own.max() == min(world, ((w + 127) // 128) * ((h + 127) // 128)) - 1
        # whole root tiles, x-major numbering
        assert (own[:128, :128] == 0).all()
        if h > 128:
            assert (own[128:256, :128] == 1 % world).all()


def _worker(rank, world, port, path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(path)                      # [H, W, 4] int32 words of the oracle's frame
    own = D.owner_map(full.shape[1], full.shape[0], D.root_tile(max(full.shape[:2])), world)
    part = torch.from_numpy(np.where((own == rank)[..., None], full, 0).astype(np.int32))
    D.combine(part, dst=0)
    ok = torch.tensor([1])
    if rank == 0:
        ok[0] = int(np.array_equal(part.numpy(), full))
    dist.broadcast(ok, src=0)
    dist.destroy_process_group()
    assert ok.item() == 1


def test_combine_two_ranks_gloo(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    O = oracle_mod
    img = O.render3d(O.Shape.from_vm(MODEL), SIZE)[0]
    words = img.view(np.int32).reshape(SIZE, SIZE, 4)
    path = str(tmp_path / "frame.npy")
    np.save(path, words)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, path), nprocs=2, join=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_device_shards_sum_to_the_frame(world):
    import fidget_amd as F
    shape = F.Shape.from_vm(MODEL)
    full = F.render3d(shape, SIZE)[0].view(np.int32).reshape(SIZE, SIZE, 4)
    own = D.owner_map(SIZE, SIZE, D.root_tile(SIZE), world)
    acc = np.zeros_like(full)
    for r in range(world):
        part = F.render3d(shape, SIZE, shard=r, n_shards=world)[0].view(np.int32).reshape(SIZE, SIZE, 4)
        assert np.array_equal(part, np.where((own == r)[..., None], full, 0)), f"shard {r} is not full * ownership mask"
        acc += part
    assert np.array_equal(acc, full)
