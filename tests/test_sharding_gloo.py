"""Multi-process path on CPU: world_size 2 and 3 over gloo.  The band geometry comes from the product library
(itwBandForPart, loadable without a GPU); the per-band encoder is injected (the oracle) because there is no GPU
here -- what is under test is the sharding/gather logic that bench.py --gpus N runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fmt, prof, h, w, q):
    for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from itw_amd import shard, surfaces
    from oracle import pyoracle
    img = surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)

    def cpu_encode(f, band, settings):            # stands in for the HIP path on this GPU-less box
        return torch.from_numpy(pyoracle.encode(f, band.numpy(), settings))

    t = torch.from_numpy(img.view(np.int16) if fmt == "bc6h" else img)
    src = torch.from_numpy(img)
    full = shard.encode_sharded(fmt, src if fmt != "bc6h" else torch.from_numpy(img), prof, encode=cpu_encode)
    want = pyoracle.encode(fmt, img, prof)
    q.put((rank, bool((full.numpy() == want).all()), int(full.numel())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fmt,prof,h,w", [(2, "bc1", None, 64, 32), (2, "bc7", "veryfast", 32, 32),
                                                (3, "bc3", None, 40, 16), (2, "bc6h", "fast", 16, 32)])
def test_band_sharding_and_gather(world, fmt, prof, h, w):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fmt, prof, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1


def _pipeline_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from itw_amd import shard
    nbytes = 4096
    state = {"step": 0}

    def encode_into(out):                          # stands in for the HIP encode: content depends on (rank, step)
        out.copy_(torch.full((nbytes,), (17 * state["step"] + 3 * rank + 1) % 256, dtype=torch.uint8))

    pipe = shard.BandPipeline(nbytes, world, rank, torch.device("cpu"), encode_into)
    ok = True
    history = []
    for step in range(7):
        state["step"] = step
        history.append((step, pipe.step()))
        if step >= 1:                              # the previous step's image is complete once its gather is waited for
            ps, pb = history[-2]
            if pipe.work[pb] is not None:
                pipe.work[pb].wait()
                pipe.work[pb] = None
            want = torch.cat([torch.full((nbytes,), (17 * ps + 3 * r + 1) % 256, dtype=torch.uint8) for r in range(world)])
            ok = ok and bool((pipe.full[pb] == want).all())
    pipe.drain()
    ps, pb = history[-1]
    want = torch.cat([torch.full((nbytes,), (17 * ps + 3 * r + 1) % 256, dtype=torch.uint8) for r in range(world)])
    ok = ok and bool((pipe.full[pb] == want).all()) and all(w is None for w in pipe.work)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gather_double_buffering(world):
    """bench.py --gpus N overlaps the all-gather of step i with the encode of step i+1 (shard.BandPipeline); here the same
    object over gloo: every step's whole image is exactly the bands of that step, buffers alternate, nothing is left in flight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _strong_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from itw_amd import shard
    geo = bench.plan("strong", 16384, world, rank, "bc7")
    bx = geo["width"] // 4
    r0, nrows = geo["y0"] // 4, geo["rows"] // 4

    def encode_into(out):            # stands in for the HIP encode: every 16-byte block carries its global block row
        rows = torch.arange(r0, r0 + nrows, dtype=torch.int32).repeat_interleave(bx * 4)      # 4 int32 per block
        out.view(torch.int32).copy_(rows)

    pipe = shard.BandPipeline(geo["band_bytes"], world, rank, torch.device("cpu"), encode_into, depth=1)
    b = pipe.step()
    pipe.drain()
    full = pipe.full[b].view(torch.int32).view(-1, bx * 4)
    ok = full.shape[0] == 4096 and bool((full[:, 0] == torch.arange(4096, dtype=torch.int32)).all()) \
        and bool((full[:, -1] == torch.arange(4096, dtype=torch.int32)).all())
    q.put((rank, ok, geo))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_sharding_of_the_16384_surface_world_2():
    """BASELINE configs[4] as bench.py --gpus N runs it by default: ONE 16384^2 BC7 surface, rank r encodes block rows
    [4096 r / N, 4096 (r+1) / N) in place at its offset of the whole-image stream, bands all-gathered.  World 2 over gloo
    with a stand-in encoder that writes each block's global row: the gathered 256 MiB stream is in raster order."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    geos = {r: g for r, _, g in res}
    assert geos[0]["rows"] == geos[1]["rows"] == 8192 and geos[1]["y0"] == 8192
    assert geos[0]["band_bytes"] * 2 == geos[0]["total_bytes"] == 16384 * 16384 // 16 * 16


def test_bench_plan_weak_and_strong_geometry():
    import bench
    for world in (1, 2, 4, 8):
        offs = []
        for r in range(world):
            g = bench.plan("strong", 16384, world, r, "bc7")
            assert g["height"] == 16384 and g["rows"] == 16384 // world and g["y0"] == r * g["rows"]
            offs.append((g["band_off"], g["band_bytes"]))
            w = bench.plan("weak", 4096, world, r, "bc7")
            assert (w["width"], w["height"], w["rows"], w["y0"]) == (4096, 4096 * world, 4096, 4096 * r)
        assert offs == [(i * offs[0][1], offs[0][1]) for i in range(world)] and offs[0][1] * world == 16384 * 16384


@pytest.mark.parametrize("fmt,h,w,parts", [("bc5", 62, 61, 3), ("bc4", 9, 5, 2), ("bc5", 64, 64, 5), ("bc4", 3, 3, 4)])
def test_band_rule_keeps_partial_blocks_for_bc4_bc5(fmt, h, w, parts):
    """BC4 / BC5 keep the partial last block row and column (itw_bc45.h), so their bands follow ceil rules
    (itwBandForPartEx): band streams of the CPU oracle concatenate to the whole-surface stream."""
    from oracle import pyoracle
    from itw_amd import shard, surfaces
    img = np.ascontiguousarray(surfaces.ldr_smooth(64, 64)[:h, :w])
    whole = pyoracle.encode(fmt, img).reshape(-1)
    got = np.zeros_like(whole)
    covered = 0
    for r in range(parts):
        y0, rows, off, nbytes = shard.band_of(w, h, fmt, r, parts)
        assert off == covered
        if rows > 0:
            got[off:off + nbytes] = pyoracle.encode(fmt, np.ascontiguousarray(img[y0:y0 + rows])).reshape(-1)
        covered += nbytes
    assert covered == whole.size and np.array_equal(got, whole)


def _run_bench_world2(extra_env, extra_args=()):
    import subprocess
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, ITW_BENCH_CONTROL_FLOW_TEST="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ITW_BENCH_DIST_TIMEOUT_S="120", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--size", "512", "--no-formats", "--no-cpu", *extra_args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    return outs


@pytest.mark.parametrize("corrupt,interleave", [(None, 1), (None, 4), ("0", 4), ("1", 4)])
def test_bench_n_gt_1_verifies_the_gathered_image(corrupt, interleave):
    """the N > 1 bench job checks what it gathered.  World 2 over gloo with the stand-in encoder (block
    bytes = a function of the band's texels): an intact run reports gather_verified = true; a rank that damages one byte of
    its band after encoding it (ITW_BENCH_CORRUPT_RANK) makes every rank's comparison of that band fail -> false, with the
    damaged bytes counted (its own check + the other rank's check of it, per job)."""
    import json
    outs = _run_bench_world2({} if corrupt is None else {"ITW_BENCH_CORRUPT_RANK": corrupt}, ("--interleave", str(interleave)))
    j = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    # world 2: own share + the other's, on both ranks; round 5: a share is `interleave` sub-bands (K * N sub-bands dealt round-robin:
    # the content-aware partition), each gathered by its own in-place all_gather
    assert j["band_checks"] == 4 * interleave and len(j["per_rank_kernel_ms"]) == 2
    assert f"{interleave} interleaved" in j["config"]["sharding"]
    if corrupt is None:
        assert j["gather_verified"] is True and j["mismatching_bytes"] == 0
    else:
        assert j["gather_verified"] is False and j["mismatching_bytes"] == 2   # seen by the damaged rank itself and by its peer


def test_bench_weak_scaling_job_verifies_too():
    """--scaling weak: every rank owns its own seeded 512 x 512 band of a 512 x 1024 surface; the verification regenerates the
    other rank's band from ITS seed.  Intact -> true; rank 1 damaged -> false."""
    import json
    for corrupt, want in ((None, True), ("1", False)):
        outs = _run_bench_world2({} if corrupt is None else {"ITW_BENCH_CORRUPT_RANK": corrupt}, ("--scaling", "weak"))
        j = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
        assert j["scaling"] == "weak" and "512x1024" in j["config"]["workload"] and j["gather_verified"] is want, j


def test_bench_n_gt_1_control_flow_runs_end_to_end_on_cpu():
    """bench.py --gpus 2 as the driver launches it (one process per rank, env rendezvous), with ITW_BENCH_CONTROL_FLOW_TEST=1:
    gloo instead of RCCL and a memset instead of the encode.  Not a measurement -- it executes every Python line of the N > 1
    path (strong plan of one surface, bands, pipelined in-place all-gather, max over ranks, weak side figure, the JSON line)."""
    import json
    import subprocess
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, ITW_BENCH_CONTROL_FLOW_TEST="1", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--size", "512", "--no-formats", "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 2 and j["config"]["ranks_seen_by_rccl"] == 2
    assert "512x512" in j["config"]["workload"] and j["weak_side"]["steps"] >= 3 and j["value"] > 0
    # what the first hardware run is read by: per-rank encode and gather times, their imbalance, K
    c = j["config"]
    assert len(c["per_rank_encode_ms"]) == 2 and len(c["per_rank_gather_ms"]) == 2 and all(v > 0 for v in c["per_rank_gather_ms"])
    assert c["max_over_mean"] >= 1.0 and c["sub_bands_per_rank"] in (1, 2, 4) and "rccl_version" in c
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]          # only rank 0 prints


@pytest.mark.parametrize("child", ["ok", "bad", "crash"])
def test_bench_n_gt_1_cpp_host_job_control_flow(child):
    """Round 4: for N > 1 bench.py runs the job a second time through the C++ host path -- rank 0 starts a child process
    (`bench.py --cpp-worker`: one process, itwCompressImageMultiGPUEx over all N GPUs) while the other ranks wait at a CPU
    (gloo) barrier -- and makes it the headline when it returns a verified image; the torch.distributed job stays as
    `python_side`.  Here (no GPU) the child prints a canned account: what is under test is the control flow -- environment
    scrubbing, barriers, the merge, and the fallback when the child reports an unverified image or dies."""
    import json
    env = {"ITW_BENCH_FAKE_CPP": {"ok": "1", "bad": "bad", "crash": "1"}[child]}
    if child == "crash":
        env["ITW_BENCH_CPP_WORKER_CRASH"] = "1"
    outs = _run_bench_world2(env)
    j = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert j["python_side"]["gather_verified"] is True and j["python_side"]["ranks_seen_by_rccl"] == 2 and j["python_side"]["value"] > 0
    if child == "ok":
        assert j["config"]["host"].startswith("cpp") and j["config"]["transport"] == "rccl" and j["config"]["ranks_seen_by_rccl"] == 2
        assert j["ms_per_step"] == 2.0 and j["gather_verified"] is True and j["band_checks"] == 2 and "error" not in j["cpp_host"]
    else:
        assert j["config"]["host"].startswith("python") and j["value"] == j["python_side"]["value"] and "note" in j["cpp_host"]
        assert ("error" in j["cpp_host"]) == (child == "crash")
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]


def _interleaved_worker(rank, world, port, pieces, q):
    """InterleavedPipeline over gloo: K sub-bands per rank, each group of `world` sub-bands gathered in place by its own all_gather; the
    stand-in encoder writes (step, sub-band index) so that every byte of every step's image is known"""
    import torch
    import torch.distributed as dist
    from itw_amd import shard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        pb = 96
        state = {"step": 0}

        def enc(k):
            def f(out):
                out.fill_((state["step"] * 16 + k * world + rank) % 251)
            return f
        pipe = shard.InterleavedPipeline(pb, world, rank, torch.device("cpu"), [enc(k) for k in range(pieces)])
        ok = True
        for step in range(5):
            state["step"] = step
            b = pipe.step()
            pipe.drain()
            want = torch.cat([torch.full((pb,), (step * 16 + j) % 251, dtype=torch.uint8) for j in range(pieces * world)])
            ok = ok and bool((pipe.full[b] == want).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,pieces", [(2, 1), (2, 4), (3, 2)])
def test_interleaved_pipeline_gathers_every_sub_band_to_its_place(world, pieces):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_interleaved_worker, args=(r, world, port, pieces, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_sub_band_geometry_is_the_band_rule_on_k_times_n_parts():
    from itw_amd import shard
    w, h, fmt, world, k = 64, 1024, "bc7", 4, 4
    seen = []
    for piece in range(k):
        for r in range(world):
            y0, rows, off, nbytes = shard.sub_band_of(w, h, fmt, r, world, piece, k)
            assert (y0, rows, off, nbytes) == shard.band_of(w, h, fmt, piece * world + r, k * world)
            seen.append((y0, rows))
    assert sorted(seen) == [(64 * j, 64) for j in range(16)]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dry_run_plan_is_the_band_rule_and_tiles_the_stream(world):
    """`bench.py --dry-run-plan N`: every rank's sub-bands, offsets and message sizes for the 16384^2 surface, without a
    GPU.  Each sub-band must be itwBandForPart(j, K * N)'s, the sub-bands must tile the block stream exactly once, and the collectives'
    buffers must be the K groups of N equal pieces the in-place all-gather needs."""
    import json
    import subprocess
    import itw_amd
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-plan", str(world)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    K, size = j["sub_bands_per_rank"], j["size"]
    assert size == 16384 and j["ranks"] == world and K == 4 and j["total_out_bytes"] == (size // 4) ** 2 * 16
    covered = []
    for rk in j["per_rank"]:
        assert len(rk["sub_bands"]) == K and rk["device"] == rk["rank"]
        for sb in rk["sub_bands"]:
            assert sb["sub_band"] % world == rk["rank"] and sb["sub_band"] // world == sb["piece"]
            y0, rows, off = itw_amd.band_for_part(size, size, "bc7", sb["sub_band"], K * world)
            assert (sb["first_texel_row"], sb["texel_rows"], sb["out_offset"]) == (y0, rows, off)
            assert sb["out_bytes"] == (rows // 4) * (size // 4) * 16 and sb["input_bytes"] == rows * size * 4
            covered.append((sb["out_offset"], sb["out_bytes"]))
        assert rk["out_bytes"] * world == j["total_out_bytes"]                    # equal load in bytes
    covered.sort()
    assert covered[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(covered, covered[1:])) and sum(c[1] for c in covered) == j["total_out_bytes"]
    piece = j["per_rank"][0]["sub_bands"][0]["out_bytes"]
    for k, c in enumerate(j["python_job_collectives"]):
        assert c["group"] == k and c["send_bytes_per_rank"] == piece and c["buffer_offset"] == k * world * piece and c["buffer_bytes"] == world * piece
        # group k's buffer is exactly sub-bands k*N .. k*N+N-1, in rank order: the in-place all-gather's layout
        for rk in j["per_rank"]:
            assert rk["sub_bands"][k]["out_offset"] == c["buffer_offset"] + rk["rank"] * piece
    assert j["cpp_job_gather"]["owner_recv_bytes_total"] == j["total_out_bytes"] - K * piece
