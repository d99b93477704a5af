"""Two ranks on the one GPU of the test box, gloo backend: rehearses libstevi_amd.sharded.stereoMatchSharded (the
code path bench.py --gpus N runs over RCCL) with real processes and a real all_reduce of the device-resident keys."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import libstevi_amd as sv
    from helpers import parallax_pair
    from libstevi_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        src, tgt, _ = parallax_pair(64, 320, 20, 10, 40, 3, 19, seed=9)
        d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
        D = 257  # odd on purpose: uneven shards
        res = sharded.stereoMatchSharded(d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=100.0, refineKernel=sv.InterpolationKernel.Parabola,
                                         refine_h_radius=4, refine_v_radius=4)
        full = sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=100.0,
                              refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=4, refine_v_radius=4)
        torch.cuda.synchronize()
        assert torch.equal(res["disp"], full["disp"])
        a, b = res["refined"].cpu().numpy(), full["refined"].cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
        # the same frames through the pipelined exchange: results one submit late, identical
        pipe = sharded.ShardedStereoPipeline(4, 4, D, sgmDirections=8, Pout=100.0)
        frames = []
        for k in range(3):
            s_k, t_k, _ = parallax_pair(64, 320, 20, 10, 30 + 5 * k, 3, 19, seed=20 + k)
            frames.append((torch.from_numpy(t_k).to(dev), torch.from_numpy(s_k).to(dev)))
        got = [pipe.submit(t_k, s_k) for t_k, s_k in frames]
        assert got[0] is None
        got = got[1:] + [pipe.flush()]
        for (t_k, s_k), r in zip(frames, got):
            ref = sv.stereoMatch(sv.matchingFunctions.CENSUS, t_k, s_k, 4, 4, D, sgmDirections=8, Pout=100.0)
            assert torch.equal(r["disp"], ref["disp"])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
