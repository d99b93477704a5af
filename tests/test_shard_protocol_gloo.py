"""The N > 1 path on CPU: two processes, gloo backend, real all_reduce(MIN) of the shard keys.  The keys come from
the numpy restatement in tests/shard_protocol.py fed by the oracle's Hamming volume; the decoded winner must equal the
oracle's census + SGM-8 + extractSelectedIndex chain on the whole disparity range."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle as so
    from helpers import parallax_pair
    from libstevi_amd.sharded import shard_range
    from shard_protocol import finish, shard_keys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for case, (D, n_dir, Pout, margins) in enumerate([(21, 8, 100.0, (0, 0, 0, 0)), (40, 8, 7.0, (2, 1, 0, 3)), (9, 4, 100.0, (0, 0, 0, 0)),
                                                          (13, 0, 100.0, (0, 0, 0, 0))]):
            src, tgt, _ = parallax_pair(19, 30, 6, 5, 8, 1, 4, seed=100 + case)
            cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)  # every rank could build only its slice; the oracle builds all
            begin, count = shard_range(D, rank, world)
            keys = torch.from_numpy(shard_keys(cv[:, :, begin:begin + count], begin, cv.shape[1]))
            dist.all_reduce(keys, op=dist.ReduceOp.MIN)
            idx = finish(keys.numpy(), n_dir, Pout, margins)
            vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, margins, Pout) if n_dir else cv
            exp = so.extract_index(vol, so.COST)
            assert np.array_equal(idx, exp), f"rank {rank} case {case}: {(idx != exp).sum()} pixels differ"
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_disparity_shards_over_gloo(tmp_path):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _pipeline_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle as so
    from helpers import parallax_pair
    from libstevi_amd.sharded import ShardedStereoPipeline
    from shard_protocol import finish, shard_keys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, n_dir, Pout, margins = 23, 8, 100.0, (0, 0, 0, 0)

        class NumpyStages(ShardedStereoPipeline):  # the pipeline's ordering and exchange, with the protocol restated in numpy
            plane0_only = False

            def _keys(self, img_l, img_r):
                cv = so.unfold_cost_volume(so.CENSUS, img_l, img_r, 4, 4, D)
                begin, count = self.shard
                return torch.from_numpy(shard_keys(cv[:, :, begin:begin + count], begin, cv.shape[1], cv if self.plane0_only else None))

            def _plane0_only(self, img_l, img_r):
                return self.plane0_only

            def _finish(self, img_l, img_r, keys):
                return finish(keys.numpy(), n_dir, Pout, margins)

        frames = [parallax_pair(17, 28, 6, 5, 3 + k, 1, 4, seed=300 + k) for k in range(4)]
        for plane0_only in (False, True):  # both planes reduced / plane 1 global as written, plane 0 alone travels
            pipe = NumpyStages(4, 4, D)
            pipe.plane0_only = plane0_only
            results = [pipe.submit(tgt, src) for src, tgt, _ in frames]
            assert results[0] is None and pipe._in_flight is not None
            results = results[1:] + [pipe.flush()]
            assert pipe.flush() is None
            for k, ((src, tgt, _), idx) in enumerate(zip(frames, results)):
                cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
                exp = so.extract_index(so.sgm(cv, n_dir, so.COST, 0.001, 0.01, margins, Pout), so.COST)
                assert np.array_equal(idx, exp), f"rank {rank} frame {k} plane0_only={plane0_only}: {(idx != exp).sum()} pixels differ"
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_pipelined_exchange_over_gloo(tmp_path):
    """ShardedStereoPipeline: async all-reduce of frame k in flight while frame k + 1 is submitted; results arrive one
    submit late, in order, each equal to the unsharded oracle chain."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _band_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle as so
    from helpers import parallax_pair
    from libstevi_amd.sharded import RowBandStereoPipeline
    from shard_protocol import band_winner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for case, (D, n_dir, Pout, margins) in enumerate([(21, 8, 100.0, (0, 0, 0, 0)), (16, 8, 7.0, (2, 1, 0, 3)), (9, 4, 100.0, (0, 0, 0, 0)),
                                                          (13, 0, 100.0, (0, 0, 0, 0))]):
            frames = [parallax_pair(23, 31, 6, 5, 3 + k, 1, 4, seed=500 + 10 * case + k) for k in range(3)]

            def compute(img_l, img_r, rows):  # the band from its own rows + halo only
                return torch.from_numpy(band_winner(img_l, img_r, 4, 4, D, rows, n_dir, Pout, margins, oracle=so))

            pipe = RowBandStereoPipeline(4, 4, D, gather=True, compute=compute, sgmDirections=n_dir, Pout=Pout)
            results = [pipe.submit(tgt, src) for src, tgt, _ in frames]
            assert results[0] is None
            results = results[1:] + [pipe.flush()]
            assert pipe.flush() is None
            for k, ((src, tgt, _), idx) in enumerate(zip(frames, results)):
                cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
                vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, margins, Pout) if n_dir else cv
                exp = so.extract_index(vol, so.COST)
                assert np.array_equal(idx.numpy(), exp), f"rank {rank} case {case} frame {k}: {(idx.numpy() != exp).sum()} pixels differ"
            # without the gather every rank keeps its own rows
            own = RowBandStereoPipeline(4, 4, D, compute=compute, sgmDirections=n_dir, Pout=Pout)
            src, tgt, _ = frames[0]
            b, c = own.rows_of(tgt, src)
            assert np.array_equal(own.submit(tgt, src).numpy(), exp_rows(so, tgt, src, D, n_dir, Pout, margins)[b:b + c])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def exp_rows(so, tgt, src, D, n_dir, Pout, margins):
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
    vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, margins, Pout) if n_dir else cv
    return so.extract_index(vol, so.COST)


@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_over_gloo(tmp_path, world):
    """RowBandStereoPipeline: every rank computes its rows of the map from its own rows + halo (numpy restatement of
    svh_census_band_match on the oracle's census volume), one all_gather in flight replicates the map; it equals the oracle's
    census + SGM + extractSelectedIndex chain on the whole image -- the rows of the map really are independent."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_band_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_range_partitions():
    from libstevi_amd.sharded import shard_range
    for total in (1, 7, 256, 257, 2048):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
