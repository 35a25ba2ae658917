"""CPU, world_size 2 over gloo: image sharding covers the batch exactly once and the single
all_gather of result records reassembles every rank's results in rank order."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glass_amd.distributed import all_gather_records, pack_results, shard_indices, unpack_results
    from glass_amd.structures.core import Instances, RotatedBoxes
    n_images = 6
    mine = shard_indices(n_images, rank, world)
    res = []
    for g in mine:                                     # deterministic fake result of image g
        k = g % 4
        r = Instances((100 + g, 200))
        r.pred_boxes = RotatedBoxes(torch.full((k, 5), float(g)))
        r.scores = torch.full((k,), 0.1 * g)
        r.pred_classes = torch.zeros((k,), dtype=torch.int64)
        r.orientations = torch.zeros((k, 2))
        r.pred_text_prob = torch.softmax(torch.full((k, 26, 97), 0.0) + torch.arange(97.0) * (g + 1) * 0.01, -1)
        res.append(r)
    rec = pack_results(res, 4, 26)
    allrec = all_gather_records(rec)
    assert allrec.shape[0] == world
    flat = allrec.reshape(-1, allrec.shape[-1])
    back = unpack_results(flat, [(100 + g, 200) for g in range(n_images)], 4, 26)
    ok = all(len(b) == g % 4 and (len(b) == 0 or float(b.pred_boxes.tensor[0, 0]) == float(g)) and
             (len(b) == 0 or int(b.pred_char_index[0, 0]) == 96) for g, b in enumerate(back))
    # without `rows`, ranks that hold different record counts must all RAISE (count exchange), not hang or mis-view
    raised = False
    try:
        all_gather_records(rec[: 2 + rank])
    except ValueError as e:
        raised = "different record counts" in str(e)
    q.put((rank, mine, ok and raised))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 1, 2] and got[1][1] == [3, 4, 5]
    assert all(g[2] for g in got)


def _fake_words(rank, step, n_images, D, T):
    """deterministic padded word-post-processor outputs of (rank, step): every field encodes its owner"""
    K = D
    tag = 100.0 * step + 10.0 * rank
    count = torch.tensor([(i + rank + step) % (D + 1) for i in range(n_images)], dtype=torch.int32)
    return {"count": count,
            "boxes": torch.full((n_images, K, 5), tag) + torch.arange(n_images).view(-1, 1, 1),
            "scores": torch.full((n_images, K), tag + 0.5),
            "text_score": torch.full((n_images, K), tag + 0.25),
            "polygons": torch.full((n_images, K, 4, 2), tag + 1.0),
            "text_len": torch.full((n_images, K), (step % 3) + 1, dtype=torch.int32),
            "char": torch.full((n_images, K, T), 3 + (step + rank) % 5, dtype=torch.int32)}


def _pipelined_worker(rank, world, port, q):
    """the bench step's host schedule (bench.py step_g) with two steps in flight: three read-back yields, pack_words,
    ONE all_gather of the ragged shards; the step's own outputs must come back from every rank, in step order."""
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glass_amd.distributed import (all_gather_records, gathered_to_global, pack_words, shard_indices, shard_rows,
                                       unpack_words, words_record_size)
    from glass_amd.utils.pipeline import ReadBack, StepOutput, run_pipelined
    N_GLOBAL, D, T, STEPS = 7, 4, 6, 5                     # 7 images over 2 ranks: shards of 3 and 4 (ragged)
    mine = shard_indices(N_GLOBAL, rank, world)
    rows = shard_rows(N_GLOBAL, world)
    chars = [chr(ord("a") + i) for i in range(26)]
    order = []

    def make(step):
        def gen():
            words = _fake_words(rank, step, len(mine), D, T)
            h = yield ReadBack(words["count"])                 # detection counts
            assert h[0].tolist() == words["count"].tolist()
            yield ReadBack(words["count"])                     # surviving counts
            out = StepOutput()
            out.words = words
            yield ReadBack(words["count"], words["char"], words["text_len"])   # word post-processor
            order.append(step)
            rec = pack_words(out.words, D, T)
            assert rec.shape == (len(mine), words_record_size(D, T))
            return step, all_gather_records(rec, rows=rows)
        return gen

    res = run_pipelined([make(s) for s in range(STEPS)], depth=2, device="cpu")
    ok = [s for s, _ in res] == list(range(STEPS)) and order == list(range(STEPS))
    for step, allrec in res:
        ok &= tuple(allrec.shape[:2]) == (world, rows)
        glob = gathered_to_global(allrec, N_GLOBAL)
        ok &= glob.shape[0] == N_GLOBAL
        back = unpack_words(glob, D, T, chars)
        for g, b in enumerate(back):
            r = 0 if g in shard_indices(N_GLOBAL, 0, world) else 1
            i = g - shard_indices(N_GLOBAL, r, world)[0]
            k = (i + r + step) % (D + 1)
            tag = 100.0 * step + 10.0 * r
            ok &= len(b["texts"]) == k
            if k:
                ok &= float(b["boxes"][0, 0]) == tag + i and abs(float(b["scores"][0]) - (tag + 0.5)) < 1e-6
                ok &= b["texts"][0] == chars[3 + (step + r) % 5] * ((step % 3) + 1)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pipelined_steps_ragged_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == [(0, True), (1, True)]


def test_gather_rejects_oversized_shard_and_pads_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
    import pytest
    from glass_amd.distributed import all_gather_records, gathered_to_global, shard_rows
    assert shard_rows(7, 2) == 4 and shard_rows(8, 8) == 1 and shard_rows(3, 8) == 1
    rec = torch.arange(6.0).view(3, 2)
    out = all_gather_records(rec, rows=4)                   # no process group: world 1
    assert out.shape == (1, 4, 2) and float(out[0, 3].abs().sum()) == 0.0
    assert torch.equal(gathered_to_global(out, 3), rec)
    with pytest.raises(ValueError):
        all_gather_records(rec, rows=2)


def test_segment_scoped_restores_between_segments():
    """process-global settings (conv precision) are switched per segment, never across a yield."""
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
    from glass_amd.utils.pipeline import ReadBack, run_pipelined, segment_scoped
    state = {"v": "fp32"}
    seen = []

    def setv(v):
        prev, state["v"] = state["v"], v
        return prev

    def body(name, want):
        for seg in range(3):
            seen.append((name, seg, state["v"]))
            assert state["v"] == want
            yield ReadBack(torch.zeros(1))
        return name

    def mk(name, want):
        return lambda: segment_scoped(body(name, want), lambda: setv(want), setv)

    res = run_pipelined([mk("a", "fp16"), mk("b", "fp32"), mk("c", "fp16")], depth=2, device="cpu")
    assert res == ["a", "b", "c"] and state["v"] == "fp32" and len(seen) == 9


# --------------------------------------------------------------------------- the rank launcher (bench.py --gpus N without torchrun)
def _pkg():
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))


def test_launcher_gives_every_rank_the_torchrun_environment(tmp_path):
    """`python bench.py --gpus N` starts its N ranks itself (reference tools/eval_glass.py:199-206 `launch`).
    Every rank must see RANK / LOCAL_RANK = its index, the same WORLD_SIZE / MASTER_PORT and MASTER_ADDR = 127.0.0.1."""
    _pkg()
    from glass_amd.distributed import launch_local_ranks, rank_env
    child = ("import os, sys; open(os.path.join(sys.argv[1], os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
             "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')))")
    rc = launch_local_ranks([sys.executable, "-c", child, str(tmp_path)], 3)
    assert rc == 0
    rows = [open(tmp_path / str(r)).read().split() for r in range(3)]
    assert [r[0] for r in rows] == ["0", "1", "2"] and [r[1] for r in rows] == ["0", "1", "2"]
    assert {r[2] for r in rows} == {"3"} and {r[3] for r in rows} == {"127.0.0.1"} and len({r[4] for r in rows}) == 1
    assert {r[5] for r in rows} == {"0"}
    e = rank_env(1, 2, 1234, base={})
    assert (e["RANK"], e["WORLD_SIZE"], e["MASTER_PORT"], e["MASTER_ADDR"]) == ("1", "2", "1234", "127.0.0.1")
    import pytest
    with pytest.raises(ValueError):
        rank_env(2, 2, 1234)


def test_launcher_propagates_a_failing_rank_and_stops_the_others(tmp_path):
    """a rank that dies must not leave the others waiting in a collective: its exit code is returned and the remaining ranks
    (here: sleeping for a minute) are terminated - only the process groups the launcher itself started"""
    _pkg()
    import time
    from glass_amd.distributed import launch_local_ranks
    child = ("import os, sys, time\n"
             "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
             "time.sleep(60)\n")
    t0 = time.time()
    rc = launch_local_ranks([sys.executable, "-c", child], 2, grace_s=3.0)
    assert rc == 7 and time.time() - t0 < 30


def test_launched_ranks_form_a_gloo_group_and_gather(tmp_path):
    """the launcher's environment is all `init_process_group` needs (env:// rendezvous): two launched ranks gather records"""
    _pkg()
    from glass_amd.distributed import launch_local_ranks
    child = f"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {os.path.join(ROOT, "glass-text-spotting_amd")!r})
from glass_amd.distributed import all_gather_records
dist.init_process_group("gloo")
r = dist.get_rank()
out = all_gather_records(torch.full((2, 3), float(r)), rows=2)
assert out.shape == (2, 2, 3) and float(out[0].sum()) == 0.0 and float(out[1].sum()) == 6.0
dist.barrier(); dist.destroy_process_group()
"""
    assert launch_local_ranks([sys.executable, "-c", child], 2) == 0


def test_bench_refuses_to_degrade_without_gpus():
    """`python bench.py --gpus 2` on a box that cannot run it exits non-zero and prints no JSON line (it used to run ONE rank
    silently when WORLD_SIZE was unset)."""
    import subprocess
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""
    # a torchrun-style environment whose WORLD_SIZE contradicts --gpus is refused as well
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""


def _oversize_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glass_amd.distributed import all_gather_records, unpack_words
    rec = torch.zeros((3 if rank == 1 else 2, 1 + 4 * (5 + 1 + 1 + 8 + 1 + 2)))
    how = None
    try:
        out = all_gather_records(rec, rows=2)            # rank 1 holds 3 > rows: must not hang rank 0 in the collective
        unpack_words(out.reshape(-1, out.shape[-1]), 4, 2, "ab")
    except ValueError as e:
        how = "local" if "local records > rows" in str(e) else ("poisoned" if "poisoned" in str(e) else str(e))
    q.put((rank, how))
    dist.barrier()
    dist.destroy_process_group()


def test_oversized_shard_with_rows_raises_on_every_rank_instead_of_hanging():
    """with `rows` set an oversized shard used to raise before the collective on its own rank only."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_oversize_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == [(0, "poisoned"), (1, "local")]


def _run_bench(extra_env, *argv, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_dry_run_two_self_launched_ranks():
    """bench.py's OWN N-rank path on CPU (GLASS_BENCH_DRYRUN=1: no GPU, no model, synthetic records): `python bench.py --gpus 2`
    launches two ranks itself, they form a gloo group, run the pipelined schedule with one all_gather of word records per step and
    rank 0 prints ONE JSON line with n_gpus == 2 == comm.world_size and a [2, B, record] gather - the plumbing the 8-GPU run uses."""
    import json
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "3")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm"]["world_size"] == 2 and d["comm"]["backend"] == "gloo" and d["data"] == "dry-run"
    assert d["comm"]["gathered_records_shape"][:2] == [2, 3] and d["comm"]["gathered_records_expected"] == 6
    assert len(d["comm"]["per_rank_ms_per_step"]) == 2
    # the same through torchrun's environment contract (the driver's N > 1 launch): a rank joins the world it is given
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # ... and refuses a world that contradicts --gpus
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--steps", "2")
    assert p.returncode != 0 and p.stdout.strip() == ""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie still answers kill(0): look at its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_ranks_never_outlive_the_launcher(tmp_path):
    """the ranks sit in their own sessions (so that the launcher can stop exactly them), which also means a signal aimed at the
    launcher does not reach them: SIGTERM to the launcher must stop them (handler -> the same shutdown path), and a launcher that
    is SIGKILLed takes them along (PR_SET_PDEATHSIG) - a timed-out bench must not leave ranks holding the GPUs."""
    import signal
    import subprocess
    import time
    child = "import os, sys, time; open(os.path.join(sys.argv[1], 'pid' + os.environ['RANK']), 'w').write(str(os.getpid())); time.sleep(120)"
    prog = (f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'glass-text-spotting_amd')!r})\n"
            f"from glass_amd.distributed import launch_local_ranks\n"
            f"sys.exit(launch_local_ranks([sys.executable, '-c', {child!r}, sys.argv[1]], 2, grace_s=3.0))\n")
    for how, sig in (("term", signal.SIGTERM), ("kill", signal.SIGKILL)):
        d = tmp_path / how
        d.mkdir()
        lp = subprocess.Popen([sys.executable, "-c", prog, str(d)])
        t_end = time.time() + 60
        while time.time() < t_end and not all((d / f"pid{r}").exists() and (d / f"pid{r}").read_text() for r in range(2)):
            time.sleep(0.1)
        pids = [int((d / f"pid{r}").read_text()) for r in range(2)]
        assert all(_alive(p) for p in pids)
        lp.send_signal(sig)
        rc = lp.wait(timeout=30)
        assert rc == (128 + signal.SIGTERM if how == "term" else -signal.SIGKILL), rc
        t_end = time.time() + 15
        while time.time() < t_end and any(_alive(p) for p in pids):
            time.sleep(0.1)
        assert not any(_alive(p) for p in pids), f"ranks survived a launcher that got {how}"


# ------------------------------------------------------------------------------------------- first-contact hardening (round 5)

def test_bench_preflight_rank_report_and_baseline_ref_in_the_dry_run():
    """`GLASS_BENCH_DRYRUN=1 python bench.py --gpus 2` prints, BEFORE any rank can hang: one preflight line (devices visible,
    RCCL version, HSA_ENABLE_IPC_MODE_LEGACY, timeout) and one line per rank (LOCAL_RANK -> device, NUMA pin report); the N > 1
    line carries the last N = 1 CPU baseline on file as `cpu_baseline_ref` and no `cpu_baseline` of its own."""
    import json
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    pre = [ln for ln in p.stderr.splitlines() if ln.startswith("[bench preflight] ")]
    assert len(pre) == 1
    rep = json.loads(pre[0][len("[bench preflight] "):])
    assert rep["world_size"] == 2 and rep["backend"] == "gloo" and "rccl_version" in rep and "HSA_ENABLE_IPC_MODE_LEGACY" in rep
    assert rep["timeout_s"] == 120.0 and "devices_visible" in rep
    ranks = sorted(ln for ln in p.stderr.splitlines() if ln.startswith("[bench rank "))
    assert len(ranks) == 2 and "LOCAL_RANK 0" in ranks[0] and "LOCAL_RANK 1" in ranks[1] and '"pinned"' in ranks[0]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert "cpu_baseline" not in d
    ref = d["cpu_baseline_ref"]
    assert ref is None or (ref["value"] > 0 and ref["source"].endswith(".json") and "cores" in ref)
    assert os.path.exists(os.path.join(ROOT, "BENCH_r04.json")) and ref is not None      # this repository has one on file


def test_a_rank_that_never_joins_makes_every_rank_exit_nonzero_within_the_timeout():
    """RCCL first contact: a rank that never reaches the rendezvous (hook: it sleeps instead) must not leave the others waiting
    for torch's default 10-30 minutes - `distributed.init_process_group` uses GLASS_DIST_TIMEOUT_S (120 s by default, 6 s here),
    the waiting rank raises, the launcher sees a non-zero exit and stops the sleeper; no line is printed."""
    import time
    t0 = time.time()
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1", "GLASS_BENCH_DRYRUN_ABSENT_RANK": "1", "GLASS_DIST_TIMEOUT_S": "6"},
                   "--gpus", "2", "--steps", "2", "--warmup", "0", timeout=180)
    dt = time.time() - t0
    assert p.returncode != 0 and p.stdout.strip() == "", (p.returncode, p.stdout)
    assert dt < 90, f"took {dt:.0f} s: the rendezvous timeout did not apply"
    assert "stopping the other ranks" in p.stderr


def test_numa_pinning_helpers(tmp_path):
    """sysfs parsing behind pin_to_gpu_numa_node: cpulist format, a fake /sys tree, the unknown-node (-1) and missing-device
    cases; without a GPU the pin is a no-op that still reports."""
    from glass_amd import distributed as D
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and D.parse_cpulist("") == [] and D.parse_cpulist("5") == [5]
    sysfs = tmp_path / "sys"
    (sysfs / "bus/pci/devices/0000:0d:00.0").mkdir(parents=True)
    (sysfs / "bus/pci/devices/0000:0d:00.0/numa_node").write_text("1\n")
    (sysfs / "bus/pci/devices/0000:1a:00.0").mkdir(parents=True)
    (sysfs / "bus/pci/devices/0000:1a:00.0/numa_node").write_text("-1\n")
    (sysfs / "devices/system/node/node1").mkdir(parents=True)
    (sysfs / "devices/system/node/node1/cpulist").write_text("64-127,192-255\n")
    node, cpus = D._numa_cpus_of_pci("0000:0d:00.0", str(sysfs))
    assert node == 1 and len(cpus) == 128 and cpus[0] == 64 and cpus[-1] == 255
    assert D._numa_cpus_of_pci("0000:1a:00.0", str(sysfs)) == (None, [])
    assert D._numa_cpus_of_pci("0000:ff:00.0", str(sysfs)) == (None, [])
    before = os.sched_getaffinity(0)
    # a torch build whose device properties lack the pci_* attributes must not be mapped to 0000:00:00.0
    import types
    from unittest import mock
    with mock.patch.object(D.torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(name="gpu")):
        assert D.gpu_numa_cpus(0) == (None, [], "?")
    with mock.patch.object(D.torch.cuda, "get_device_properties",
                           lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x0d, pci_device_id=0)):
        assert D.gpu_numa_cpus(0)[2] == "0000:0d:00.0"
    rep = D.pin_to_gpu_numa_node(0)                      # no GPU here: nothing to pin to
    assert rep["pinned"] is False and os.sched_getaffinity(0) == before and rep["cpus_after"] == len(before)
    pre = D.preflight_report(8, "nccl")
    assert pre["world_size"] == 8 and pre["devices_visible"] == 0 and pre["timeout_s"] == 120.0


def test_stdout_carries_one_line_even_when_native_libraries_write_to_it():
    """RCCL prints a version banner on stdout when a communicator is created (seen on MI355X with a one-rank nccl group,
    profiles/r05_bench_rccl_world1.json: five extra lines), gloo its connection chatter: bench.py points file descriptor 1 at
    stderr for the whole run and writes the ONE JSON line to the saved descriptor - with noise injected on fd 1 in every rank."""
    import json
    p = _run_bench({"GLASS_BENCH_DRYRUN": "1", "GLASS_BENCH_STDOUT_NOISE": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, p.stdout
    assert p.stderr.count("noise from a native library") == 2
