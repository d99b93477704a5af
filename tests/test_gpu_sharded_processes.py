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


def _rccl_worker(rank, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import libstevi_amd as sv
    from helpers import parallax_pair
    from libstevi_amd import sharded

    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    comm = sharded.RcclCommunicator(rank=0, world=1)  # a real RCCL communicator (one rank: the one GPU of the box)
    try:
        for W, D, lower in ((320, 96, 0), (300, 64, 7)):  # second plane global / both planes travel
            src, tgt, _ = parallax_pair(48, W, 20, 10, 40, 3, 19, seed=W)
            d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
            rng_ = D if lower == 0 else sv.searchOffset1(lower, lower + D - 1)
            keys = sv.censusShardKeys(d_tgt, d_src, 4, 4, rng_, (0, D), sgmDirections=8, Pout=100.0)
            before = keys.clone()
            plane0 = sv.censusShardRegion1IsGlobal(d_tgt, d_src, rng_)
            sharded.exchange_keys_rccl(keys, plane0, comm)  # svh_census_exchange_keys: MIN over one rank is the identity
            torch.cuda.synchronize()
            assert torch.equal(keys, before)
            res = sv.censusShardFinish(d_tgt, d_src, keys, 4, 4, rng_, sgmDirections=8, Pout=100.0)
            full = sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, 4, 4, rng_, sgmDirections=8, Pout=100.0)
            assert torch.equal(res["disp"], full["disp"])
        # refusals: host keys, a missing communicator
        with pytest.raises(sv._capi.SvhError):
            sharded.exchange_keys_rccl(keys[:, ::2, :], True, comm)
        open(os.path.join(out_dir, "rccl_ok"), "w").write("ok")
    finally:
        comm.destroy()


def test_exchange_through_the_c_abi_on_a_real_rccl_communicator(tmp_path):
    """svh_census_exchange_keys (what the C++ host calls) driven from Python: communicator made through ctypes with the RCCL PyTorch
    loaded, the all-reduce enqueued by the library on the context's stream.  In a process of its own: RCCL stays out of the test runner."""
    mp.spawn(_rccl_worker, args=(str(tmp_path),), nprocs=1, join=True)
    assert (tmp_path / "rccl_ok").exists()
