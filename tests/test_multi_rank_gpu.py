"""Two ranks on ONE GPU (gloo for the collective, both processes on cuda:0): the N>1 X-engine path end to end on the device --
mi355_pack3d_dev packs the send blocks, the all-to-all delivers [group][window][t][station in group][channel slice], the fused kernel
reads that receive buffer IN PLACE (stations_per_group) and correlates the windows of an exchange in one launch
(mi355_xengine_xcorrelate_n_dev); every rank's channel slab is compared bit for bit with the oracle's result for the full array, and
rank 0 gathers the slabs into the reference's [chan][baseline] matrix.  What it cannot cover is RCCL itself (one GPU per box here):
under nccl only the collective call differs (gr-clenabled_amd/shard.py).  Reference: the antenna-group / channel partition of
SURVEY 8e; per-window correlation lib/clXEngine_impl.cc:708-817."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT
from test_multi_gpu_cpu import _free_port

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %(root)r)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    o = entry.load_oracle()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, F, T, npol, windows = %(N)d, %(F)d, %(T)d, %(npol)d, %(windows)d
    rng = np.random.default_rng(77)                      # the same array on every rank
    wins = rng.integers(-128, 128, size=(2, windows, T, N, F, npol, 2), dtype=np.int64).astype(np.int8)
    Fw, Ng = F // world, N // world
    xe = pkg.clXEngine(1, 2, 0, 0, False, pkg.DTYPE_BYTE, npol, N, 1, 0, Fw, T, [])
    per = xe.get_output_buffer_size()
    ct = pkg.shard.XEngineCornerTurn(N, F, T, npol, block=xe, windows=windows)
    g0, g1 = ct.groups[rank]
    f0, f1 = ct.output_slice()
    loc = [torch.from_numpy(np.ascontiguousarray(wins[e][:, :, g0:g1] if windows > 1 else wins[e][0][:, g0:g1])).cuda() for e in range(2)]
    vis = [torch.zeros(windows * per, 2, device="cuda") for _ in range(2)]
    # two exchanges in flight on alternating slots, each correlated in one launch straight from the receive buffer
    h = ct.start(loc[0], 0)
    for e in range(2):
        nxt = ct.start(loc[1], 1) if e == 0 else None
        recv = ct.finish(h)
        if windows > 1:
            xe.xcorrelate_n_device(windows, recv, vis[e], stations_per_group=Ng)
        else:
            xe.xcorrelate_device(recv, vis[e], stations_per_group=Ng)
        h = nxt
    torch.cuda.synchronize()
    nb = N * (N + 1) // 2 * npol * npol
    for e in range(2):
        got = vis[e].cpu().numpy().view(np.complex64).reshape(windows, Fw, nb)
        for w in range(windows):
            ref = o.xengine_ichar(N, F, npol, T, wins[e][w].reshape(-1), exact=True).reshape(F, nb)
            assert np.array_equal(got[w], ref[f0:f1]), (e, w)
    # the reference's full matrix of the last window on rank 0 (rows in channel order)
    mine = vis[1].cpu().view(windows, Fw * nb * 2)[windows - 1].contiguous()
    rows = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, rows, dst=0)
    if rank == 0:
        full = torch.cat(rows).numpy().view(np.complex64)
        assert np.array_equal(full, o.xengine_ichar(N, F, npol, T, wins[1][windows - 1].reshape(-1), exact=True))
    assert pkg.shard.max_over_ranks(1.0 + rank) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
''')


@pytest.mark.parametrize("N,F,T,npol,windows,world", [(64, 256, 128, 1, 3, 2), (32, 128, 96, 2, 2, 2), (64, 512, 64, 1, 1, 4), (64, 1024, 64, 1, 8, 8)])
def test_ranks_sharing_one_gpu(gpu, oracle, tmp_path, N, F, T, npol, windows, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "N": N, "F": F, "T": T, "npol": npol, "windows": windows})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for k in range(world):
        assert "rank %d ok" % k in r.stdout


def test_bench_two_rank_dry_run(gpu, tmp_path):
    """bench.py's N>1 control flow with two ranks on the one GPU (gloo, MI355_BENCH_ONE_DEVICE): every rank reaches every collective
    (barriers, max-over-ranks, the per-GPU gathers, the X-engine exchange), rank 0 prints ONE JSON line, no secondary leg reports an
    error.  The numbers of such a run mean nothing (two ranks share a device); under the driver the same code runs over RCCL."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-sustained"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MI355_BENCH_BACKEND="gloo", MI355_BENCH_ONE_DEVICE="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
    b = d["blocks"]
    assert "error" not in b, b.get("error")
    assert "error" not in b["clXEngine_sharded"] and b["clXEngine_sharded"]["n_gpus"] == 2 and b["clXEngine_sharded"]["channels_per_rank"] == 512
    assert b["clXEngine_channel_sharded"]["channels_per_rank"] == 512 and b["clXEngine_channel_sharded"]["us_per_window_all_ranks"] > 0
    assert len(b["clPolyphaseChannelizer_64x32_stream"]["per_gpu_MSamples_per_s"]) == 2
    # the keys the driver's N = 1, 2, 4, 8 runs are read by: whole-job value, per-GPU rates, and for config 5 both sharded forms with their
    # own scaling efficiency against an N = 1 time measured in the same line
    assert d["per_gpu_MSamples_per_s"] > 0 and d["config"]["parallelism"] == "replica-per-gpu x2"
    n1 = b["clXEngine_n1_reference"]["us_per_integration_one_gpu"]
    for key in ("clXEngine_sharded", "clXEngine_channel_sharded"):
        assert b[key]["n1_us_per_integration_single_call"] == n1, key
        assert [k for k in b[key] if "efficiency" in k] == ["scaling_efficiency_vs_n1_batched"], key  # ONE efficiency figure, like for like
    assert b["clXEngine_sharded"]["alltoall_us_per_exchange"] > 0 and b["clXEngine_sharded"]["alltoall_bytes_per_link_per_exchange"] == 8 * 1024 * 32 * 512 * 2
    assert b["clXEngine_sharded"]["bound"] in ("exchange", "correlation")
    n1b = b["clXEngine_n1_reference"]["us_per_window_one_gpu_8_windows_per_launch"]  # like for like: eight windows per launch on one GPU
    for key, t in (("clXEngine_sharded", "us_per_integration"), ("clXEngine_channel_sharded", "us_per_window_all_ranks")):
        assert b[key]["n1_us_per_window_batched"] == n1b and abs(b[key]["scaling_efficiency_vs_n1_batched"] - n1b / (2 * b[key][t])) < 2e-3, key
    assert b["clXEngine_channel_sharded"]["windows_per_launch"] in (8, 16, 32)
