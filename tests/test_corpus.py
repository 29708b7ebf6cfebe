"""Host logic of the multi-GPU path: the corpus plan every rank derives on its own, and the
two-scalar reporting exchange, run as two gloo ranks on the CPU."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tts_amd import corpus


def test_plan_covers_every_utterance_once_for_any_world():
    lengths = corpus.corpus_lengths(1000, seed=1234)
    assert lengths.min() >= 20 and lengths.max() <= 200
    batches = corpus.plan_batches(lengths, 64)
    assert sorted(np.concatenate(batches).tolist()) == list(range(1000))
    assert all(len(b) == 64 for b in batches[:-1]) and len(batches[-1]) == 1000 - 64 * 15
    flat = lengths[np.concatenate(batches)]
    assert (np.diff(flat) <= 0).all()                      # longest first
    for world in (1, 2, 3, 8):
        for policy in ("round_robin", "lpt"):
            got = [corpus.rank_batches(lengths, 64, world, r, policy) for r in range(world)]
            seen = sorted(int(i) for rb in got for b in rb for i in b)
            assert seen == list(range(1000)), (world, policy)


def test_lpt_balances_better_than_round_robin_on_sorted_batches():
    lengths = corpus.corpus_lengths(4096, seed=1234)
    batches = corpus.plan_batches(lengths, 64)
    cost = np.array([len(b) * lengths[b].max() for b in batches])

    def spread(policy):
        loads = [cost[ix].sum() for ix in corpus.deal(batches, lengths, 8, policy)]
        return max(loads) / (sum(loads) / 8)
    assert spread("lpt") <= spread("round_robin")
    assert spread("lpt") < 1.02


def test_pad_batch_and_bad_arguments():
    texts = [[5, 6, 7], [8], [9, 10]]
    ids, lens = corpus.pad_batch(texts, [2, 0])
    assert ids.tolist() == [[9, 10, 0], [5, 6, 7]] and lens.tolist() == [2, 3]
    with pytest.raises(ValueError):
        corpus.deal([np.array([0])], np.array([3]), 0)
    with pytest.raises(ValueError):
        corpus.deal([np.array([0])], np.array([3]), 2, "random")


def test_single_process_timing_needs_no_process_group():
    calls = []
    dt = corpus.timed_steps(lambda: calls.append(1), steps=4, warmup=2)
    assert len(calls) == 6 and dt >= 0.0
    assert corpus.reduce_scalar(3.5, "sum") == 3.5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = corpus.corpus_lengths(300, seed=7)
        mine = corpus.rank_batches(lengths, 32, world, rank)
        frames = float(sum(int(lengths[b].sum()) for b in mine))
        # rank 1 is the slow one: the reported time must be its time on both ranks
        dt = corpus.timed_steps(lambda: time.sleep(0.02 * (rank + 1)), steps=3, warmup=1, dist=dist)
        total = corpus.reduce_scalar(frames, "sum", dist)
        np.save(os.path.join(out_dir, "r%d.npy" % rank),
                np.array([dt, total, frames] + sorted(int(i) for b in mine for i in b), dtype=np.float64))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_share_a_plan_and_agree_on_max_time_and_total(tmp_path):
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), "r%d.npy" % r)) for r in range(2))
    assert r0[0] == r1[0] and r0[0] >= 3 * 0.04                      # max over ranks, same on both
    lengths = corpus.corpus_lengths(300, seed=7)
    assert r0[1] == r1[1] == float(lengths.sum())                     # sum over ranks
    assert r0[2] + r1[2] == r0[1]
    ids0, ids1 = set(r0[3:].astype(int)), set(r1[3:].astype(int))
    assert not (ids0 & ids1) and ids0 | ids1 == set(range(300))      # disjoint, complete


def _bench(*args):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          env=env, timeout=300), json


def test_bench_launcher_starts_n_ranks_with_distinct_shards():
    """`bench.py --gpus 2 --dry-run` (no GPU, gloo): the launcher starts two ranks, each derives its share of
    the 4096-utterance corpus, and the shares are disjoint and complete."""
    run, json = _bench("--gpus", "2", "--dry-run", "--workload", "corpus")
    assert run.returncode == 0, run.stderr
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and len(line["ranks"]) == 2
    r0, r1 = line["ranks"]
    assert (r0["rank"], r1["rank"]) == (0, 1)
    assert r0["utterances"] + r1["utterances"] == 4096 and r0["batches"] + r1["batches"] == 16   # batches of 256
    assert r0["utt_checksum"] != r1["utt_checksum"]
    lengths = corpus.corpus_lengths(4096, seed=1234)
    assert line["total_frames"] == float(np.minimum(3000, 6 * lengths).sum())
    assert abs(r0["frames"] - r1["frames"]) < 0.02 * line["total_frames"]        # LPT balance
    # the same plan, derived here
    batches = corpus.plan_batches(lengths, 256)
    dealt = corpus.deal(batches, lengths, 2, "lpt")
    for r, info in enumerate((r0, r1)):
        utts = sorted(int(i) for b in dealt[r] for i in batches[b])
        assert (utts[0], utts[-1], len(utts)) == (info["first_utt"], info["last_utt"], info["utterances"])


def test_bench_refuses_more_ranks_than_gpus_and_a_mismatched_world():
    """Asking for more GPUs than the node shows fails loudly instead of running fewer ranks; so does a
    --gpus that contradicts WORLD_SIZE."""
    import torch
    run, _ = _bench("--gpus", str(torch.cuda.device_count() + 1))
    assert run.returncode != 0 and "GPU" in run.stderr
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True,
                         text=True, env=env, timeout=120)
    assert run.returncode != 0 and "WORLD_SIZE" in run.stderr


def test_corpus_texts_do_not_depend_on_the_cut():
    lengths = corpus.corpus_lengths(50, seed=3)
    texts = corpus.corpus_texts(lengths, seed=3)
    assert [len(t) for t in texts] == lengths.tolist()
    assert all(t.min() >= 3 and t.max() < 129 for t in texts)
    again = corpus.corpus_texts(lengths, seed=3)
    assert all((a == b).all() for a, b in zip(texts, again))


def test_launcher_parent_counts_gpus_from_sysfs_and_never_asks_the_runtime(tmp_path, monkeypatch):
    """The process that starts the ranks must not open the GPU (a process that has and then starts children is
    refused on the GPU pool): its device count comes from the driver's sysfs topology, not from torch.cuda."""
    import argparse
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    for node, simds in (("0", 0), ("1", 0), ("2", 1024), ("3", 1024)):
        d = tmp_path / node
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\n" % (0 if simds else 64, simds))
    assert bench.kfd_gpu_nodes(str(tmp_path)) == 2
    assert bench.kfd_gpu_nodes(str(tmp_path / "absent")) is None

    def forbidden(*a, **k):
        raise AssertionError("the launcher parent asked the GPU runtime")
    for name in ("device_count", "is_available", "init", "set_device", "current_device"):
        monkeypatch.setattr(torch.cuda, name, forbidden)
    started = []

    class FakeProc:
        def __init__(self, argv, env):
            started.append((argv, env["RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]))
            self.rank = int(env["RANK"])

        def poll(self):
            return 3 if self.rank == 1 else None if self.rank == 0 and not getattr(self, "ended", False) else -15

        def terminate(self):
            self.ended = True
    monkeypatch.setattr(bench.subprocess, "Popen", lambda argv, env: FakeProc(argv, env))
    monkeypatch.setattr(bench, "kfd_gpu_nodes", lambda: 8)
    rc = bench.launch_ranks(argparse.Namespace(gpus=2, dry_run=False), ["--gpus", "2"])
    assert rc == 3 and [s[1:] for s in started] == [("0", "2", "127.0.0.1"), ("1", "2", "127.0.0.1")]
    monkeypatch.setattr(bench, "kfd_gpu_nodes", lambda: 1)
    assert bench.launch_ranks(argparse.Namespace(gpus=2, dry_run=False), []) == 2 and len(started) == 2
