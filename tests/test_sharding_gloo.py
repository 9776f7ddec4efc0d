"""CPU, world_size 2, gloo: the N>1 path of the Monte-Carlo loop.

Frames are sharded by contiguous global range, seeds derive from the GLOBAL frame index, and the only cross-rank
step is the sum of the counters.  No GPU here, so each rank's shard is decoded by the oracle (test infrastructure)
with exactly the reference's per-frame seeding; the sharded total must equal the single-process run."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_shard(o, Hm, cws, snr, lo, cnt, max_iter):
    """experiment.h:80-123 for global frames [lo, lo+cnt): frame g <- mt19937(g+1), codeword cws[g % len]"""
    n = Hm.shape[1]
    idx = (np.arange(lo, lo + cnt)) % len(cws)
    y = np.stack([o.transmit(lo + i + 1, snr, cws[idx[i]]) for i in range(cnt)]) if cnt else np.zeros((0, n))
    bits, ok, iters = o.bp_decode(Hm, y, snr, max_iter, threads=2)
    sent = cws[idx]
    good = (ok == 1) & (bits == sent).all(axis=1)
    pseudo = (ok == 1) & ~good
    ham = ((sent == 0) & (y <= 0)).sum(axis=1) + ((sent == 1) & (y > 0)).sum(axis=1)
    return np.array([good.sum(), pseudo.sum(), cnt, ham.sum(), ham[good].sum(), ham[~good].sum(), iters.sum()], np.int64)


def _install_oracle_backed_run_experiment(E, o, Hm):
    """No GPU here: the C-ABI call behind run_experiment is replaced by the oracle decoding exactly the frames the
    C ABI would have been asked for (global range, reference seeding).  Everything above it — shard ranges, threads,
    the all_reduce, the merge — is the product's own code."""
    def fake_run(decoder, codewords, H, snr, frames=None, first_frame=0, noise="host", seed=1):
        v = _oracle_shard(o, Hm, codewords, snr, int(first_frame), int(frames), decoder)
        return E.ExperimentResult.from_vector(v)
    E.run_experiment = fake_run


def _worker(rank, world, port, frames, snr, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import acg_alp_ldpc_amd.experiment as E
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    Hm = o.read_pcm(os.path.join(ROOT, "data", "H.txt"))
    G, _ = o.get_orthogonal(Hm)
    cws = o.gen_codewords(G, 239239239, 97)
    _install_oracle_backed_run_experiment(E, o, Hm)
    # the product's N>1 path: this rank's contiguous global range + the SUM of the seven counters over gloo
    local, total = E.run_experiment_sharded(20, cws, None, snr, frames, rank=rank, world=world, noise="host")
    lo, cnt = E.shard_range(frames, rank, world)
    assert local.total == cnt
    # timing contract of bench.py: barrier, then MAX over ranks (CPU tensors on gloo — the control plane of bench.py)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, np.concatenate([total.as_vector(), [int(t.item())]]))
    dist.destroy_process_group()


def _setup():
    from oracle.pyoracle import Oracle
    o = Oracle()
    Hm = o.read_pcm(os.path.join(ROOT, "data", "H.txt"))
    G, _ = o.get_orthogonal(Hm)
    return o, Hm, o.gen_codewords(G, 239239239, 97)


@pytest.mark.parametrize("frames", [301])
def test_two_rank_shards_equal_single_process(tmp_path, frames):
    """run_experiment_sharded under 2 gloo ranks == one process over all frames (counters incl. sum of exit iterations)"""
    o, Hm, cws = _setup()
    snr = 0.5
    single = _oracle_shard(o, Hm, cws, snr, 0, frames, 20)
    out = str(tmp_path / "r.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, frames, snr, out), nprocs=2, join=True)
    got = np.load(out)
    assert (got[:7] == single).all(), (got, single)
    assert got[7] == 2  # MAX over ranks
    assert single[2] == frames and single[3] == single[4] + single[5]


@pytest.mark.parametrize("world", [1, 3, 8])
def test_inproc_shards_equal_single_process(world):
    """run_experiment_inproc (one process, one host thread per device, no torch.distributed): same merged counters as
    N = 1 for any device count, each shard asked for its contiguous global range"""
    import acg_alp_ldpc_amd.experiment as E
    o, Hm, cws = _setup()
    frames, snr = 203, 0.5
    single = _oracle_shard(o, Hm, cws, snr, 0, frames, 20)
    old = E.run_experiment
    _install_oracle_backed_run_experiment(E, o, Hm)
    try:
        made = []
        locals_, total = E.run_experiment_inproc(lambda dev: made.append(dev) or 20, cws, None, snr, frames,
                                                 devices=range(world), noise="host")
    finally:
        E.run_experiment = old
    assert sorted(made) == list(range(world))
    assert (total.as_vector() == single).all(), (total, single)
    assert [r.total for r in locals_] == [E.shard_range(frames, g, world)[1] for g in range(world)]
    assert sum(r.total for r in locals_) == frames


def test_shard_ranges_partition_the_batch():
    from acg_alp_ldpc_amd.experiment import shard_range
    for frames in (0, 1, 7, 1000, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                lo, cnt = shard_range(frames, r, world)
                assert lo == nxt and cnt >= 0
                nxt = lo + cnt
            assert nxt == frames
