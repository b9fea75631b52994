"""CPU, world_size 2 over gloo: the samples-per-pixel sharding plan and the sum-reduction that bench.py
runs over RCCL. The per-rank renderer here is the CPU oracle (no GPU in this container); what is under
test is the plan: disjoint global sample indices per (step, rank), one reduce to rank 0, same image as
one process rendering all the samples."""
import importlib
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rta = importlib.import_module("ray-tracing_amd")
    scenes = importlib.import_module("ray-tracing_amd.scenes")
    par = importlib.import_module("ray-tracing_amd.parallel")
    from oracle import bindings as ob
    W, steps, spp_step = 24, 2, 3
    s, cam, flags = scenes.sphere_room(W)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    acc = np.zeros((W, W, 3))
    for step in range(steps):
        off = par.shard_offset(step, world, rank, spp_step)
        orc.render(cam, rta.make_params(W, W, spp_step, spp_offset=off), accum=acc, n_threads=1)
    t = torch.from_numpy(acc)
    par.reduce_to_root(t, world)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_spp_sharding_two_ranks(tmp_path, rta, scenes, ob):
    import torch.multiprocessing as mp
    par = importlib.import_module("ray-tracing_amd.parallel")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "reduced.npy"))
    W, steps, spp_step = 24, 2, 3
    s, cam, flags = scenes.sphere_room(W)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    total = par.total_spp(steps, 2, spp_step)
    want = orc.render(cam, rta.make_params(W, W, total), n_threads=2)
    # same samples, different order of the final additions
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    assert np.array_equal(ob.resolve(got, total), ob.resolve(want, total))


def test_shard_offsets_are_disjoint():
    par = importlib.import_module("ray-tracing_amd.parallel")
    for world in (1, 2, 4, 8):
        seen = set()
        for step in range(5):
            for rank in range(world):
                off = par.shard_offset(step, world, rank, 16)
                block = set(range(off, off + 16))
                assert not (seen & block)
                seen |= block
        assert seen == set(range(par.total_spp(5, world, 16)))


def test_split_samples_covers_the_image():
    par = importlib.import_module("ray-tracing_amd.parallel")
    for total in (1, 7, 100, 256, 1000):
        for world in (1, 2, 3, 4, 8):
            got = []
            for rank in range(world):
                lo, hi = par.split_samples(total, rank, world)
                assert 0 <= lo <= hi <= total and hi - lo <= -(-total // world)
                got += list(range(lo, hi))
            assert got == list(range(total))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_bench_script", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_starts_its_own_ranks(monkeypatch):
    """`bench.py --gpus N` without a launcher starts torch.distributed.run on itself, before torch is imported, and passes
    the ranks' exit code on; under a launcher (WORLD_SIZE set) it does not spawn again."""
    bench = _load_bench()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd):
        seen["cmd"] = cmd
        return Done()

    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"], run=fake_run, port=29123)
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]

    # main(): spawns when no launcher set WORLD_SIZE, exits with the children's code
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv: calls.append((n, list(argv))) or 3)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    try:
        bench.main()
        assert False, "main() should have exited"
    except SystemExit as e:
        assert e.code == 3
    assert calls == [(2, ["--gpus", "2", "--steps", "1"])]
    # a launcher that started a different number of ranks is an error, not a silent mismatch
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    calls.clear()
    try:
        bench.main()
        assert False
    except SystemExit as e:
        assert e.code == 2
    assert calls == []
