"""Image sharding + result gather for multi-GPU inference (one process per GPU).

The reference's only inference-time collective is `comm.gather(self._predictions, dst=0)` of pickled
per-image results (reference glass/evaluation/text_evaluator.py:246-249).  Images are independent
(SURVEY.md §8e), so ranks take disjoint image shards with no data-path collective; results are
exchanged as ONE fixed-size record per image through a single `all_gather` (RCCL over xGMI on GPU,
gloo in the CPU tests).  Record layout (float32, `MAX_DET` = TEST.DETECTIONS_PER_IMAGE slots):
  [0]                       count
  [1 : 1+5D]                boxes (cx,cy,w,h,angle)
  [.. +D]                   scores
  [.. +D]                   classes
  [.. +2D]                  orientations (argmax, prob)
  [.. +D*T]                 argmax character index per step      (reduced text payload)
  [.. +D*T]                 its probability
Full `[D,T,97]` probability tensors stay on the owning rank (1 MB/image; gather on request with
`full_text_prob=True`).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .structures.core import Instances, RotatedBoxes


def shard_indices(num_items: int, rank: int, world_size: int) -> List[int]:
    """Contiguous block partition: item i -> rank floor(i * W / N); the union over ranks is
    exactly range(num_items) with no overlap."""
    lo = (num_items * rank) // world_size
    hi = (num_items * (rank + 1)) // world_size
    return list(range(lo, hi))


def record_size(max_det: int, steps: int, classes: int = 0) -> int:
    return 1 + max_det * (5 + 1 + 1 + 2 + 2 * steps + steps * classes)


def pack_results(results: Sequence[Instances], max_det: int, steps: int, full_text_prob: bool = False,
                 classes: int = 0) -> torch.Tensor:
    """list[Instances] -> [n_images, record] float32 on the instances' device."""
    dev = results[0].pred_boxes.tensor.device if len(results) else torch.device("cpu")
    C = classes if full_text_prob else 0
    rec = torch.zeros((len(results), record_size(max_det, steps, C)), dtype=torch.float32, device=dev)
    D = max_det
    for i, r in enumerate(results):
        k = min(len(r), D)
        rec[i, 0] = k
        if k == 0:
            continue
        o = 1
        rec[i, o:o + 5 * D].view(D, 5)[:k] = r.pred_boxes.tensor[:k]; o += 5 * D
        rec[i, o:o + D][:k] = r.scores[:k]; o += D
        rec[i, o:o + D][:k] = r.pred_classes[:k].float(); o += D
        if r.has("orientations"):
            rec[i, o:o + 2 * D].view(D, 2)[:k] = r.orientations[:k]
        o += 2 * D
        if r.has("pred_text_prob"):
            p, idx = r.pred_text_prob[:k].max(dim=2)
            rec[i, o:o + D * steps].view(D, steps)[:k] = idx.float()
            rec[i, o + D * steps:o + 2 * D * steps].view(D, steps)[:k] = p
            if C:
                rec[i, o + 2 * D * steps:].view(D, steps, C)[:k] = r.pred_text_prob[:k]
        o += 2 * D * steps
    return rec


def pack_padded(det, max_det: int, steps: int) -> torch.Tensor:
    """Same record layout as `pack_results`, built from the padded device tensors of a step
    (`model.last_batch`) with a handful of launches instead of ~12 per image."""
    N, K = det.scores.shape
    dev = det.scores.device
    D = max_det
    k = min(K, D)
    rec = torch.zeros((N, record_size(D, steps)), dtype=torch.float32, device=dev)
    rec[:, 0] = det.counts_dev.clamp(max=D).float()
    o = 1
    rec[:, o:o + 5 * D].view(N, D, 5)[:, :k] = det.boxes[:, :k]; o += 5 * D
    rec[:, o:o + D][:, :k] = det.scores[:, :k]; o += D
    o += D                                                        # classes: all zero (single 'word' class)
    if det.orient is not None:
        rec[:, o:o + 2 * D].view(N, D, 2)[:, :k] = det.orient[:, :k]
    o += 2 * D
    if det.text is not None and det.text.dim() == 4:
        p, idx = det.text[:, :k].max(dim=3)
        rec[:, o:o + D * steps].view(N, D, steps)[:, :k] = idx.float()
        rec[:, o + D * steps:o + 2 * D * steps].view(N, D, steps)[:, :k] = p
    return rec


def words_record_size(max_det: int, steps: int) -> int:
    return 1 + max_det * (5 + 1 + 1 + 8 + 1 + steps)


def pack_words(words: dict, max_det: int, steps: int) -> torch.Tensor:
    """Record of the POST-PROCESSED words of a step (the padded outputs of ops.native.postprocess_words):
    [count | boxes 5D | score D | text score D | polygon 8D | text length D | character index D*T].  Device tensors: ONE
    launch (glass_pack_word_records); host tensors (the CPU tests of the gather plumbing): the same layout with torch."""
    N, K = words["scores"].shape
    dev = words["scores"].device
    D = max_det
    if dev.type == "cuda":
        from .ops import native as KN
        return KN.pack_word_records(words, D, steps)
    k = min(K, D)
    T = min(steps, words["char"].shape[2])
    rec = torch.zeros((N, words_record_size(D, steps)), dtype=torch.float32, device=dev)
    rec[:, 0] = words["count"].clamp(max=D).float()
    o = 1
    rec[:, o:o + 5 * D].view(N, D, 5)[:, :k] = words["boxes"][:, :k]; o += 5 * D
    rec[:, o:o + D][:, :k] = words["scores"][:, :k]; o += D
    rec[:, o:o + D][:, :k] = words["text_score"][:, :k]; o += D
    rec[:, o:o + 8 * D].view(N, D, 8)[:, :k] = words["polygons"][:, :k].reshape(N, k, 8); o += 8 * D
    rec[:, o:o + D][:, :k] = words["text_len"][:, :k].float(); o += D
    rec[:, o:o + D * steps].view(N, D, steps)[:, :k, :T] = words["char"][:, :k, :T].float()
    return rec


def unpack_words(rec: torch.Tensor, max_det: int, steps: int, characters: Sequence[str]) -> List[dict]:
    """-> per image {boxes, scores, text_scores, polygons, texts}"""
    out = []
    D = max_det
    for i in range(rec.shape[0]):
        k = _record_count(rec, i)
        o = 1
        boxes = rec[i, o:o + 5 * D].view(D, 5)[:k]; o += 5 * D
        scores = rec[i, o:o + D][:k]; o += D
        tscores = rec[i, o:o + D][:k]; o += D
        polys = rec[i, o:o + 8 * D].view(D, 4, 2)[:k]; o += 8 * D
        lens = rec[i, o:o + D][:k].long().tolist(); o += D
        chars = rec[i, o:o + D * steps].view(D, steps)[:k].long().tolist()
        texts = ["".join(characters[c] for c in chars[j][:lens[j]]) for j in range(k)]
        out.append({"boxes": boxes, "scores": scores, "text_scores": tscores, "polygons": polys, "texts": texts})
    return out


def unpack_results(rec: torch.Tensor, image_sizes: Sequence[Tuple[int, int]], max_det: int, steps: int,
                   classes: int = 0) -> List[Instances]:
    out = []
    D = max_det
    for i, size in enumerate(image_sizes):
        k = _record_count(rec, i)
        o = 1
        r = Instances(tuple(size))
        r.pred_boxes = RotatedBoxes(rec[i, o:o + 5 * D].view(D, 5)[:k].clone()); o += 5 * D
        r.scores = rec[i, o:o + D][:k].clone(); o += D
        r.pred_classes = rec[i, o:o + D][:k].long(); o += D
        r.orientations = rec[i, o:o + 2 * D].view(D, 2)[:k].clone(); o += 2 * D
        r.pred_char_index = rec[i, o:o + D * steps].view(D, steps)[:k].long()
        r.pred_char_prob = rec[i, o + D * steps:o + 2 * D * steps].view(D, steps)[:k].clone()
        o += 2 * D * steps
        if classes:
            r.pred_text_prob = rec[i, o:].view(D, steps, classes)[:k].clone()
        out.append(r)
    return out


def shard_rows(num_items: int, world_size: int) -> int:
    """rows every rank contributes to `all_gather_records` for `num_items` items: the largest shard of
    `shard_indices` (= ceil(N / W)); smaller shards are padded with count-0 records."""
    return max(len(shard_indices(num_items, r, world_size)) for r in range(world_size)) if world_size > 0 else 0


def gathered_to_global(allrec: torch.Tensor, num_items: int) -> torch.Tensor:
    """[world, rows, record] from `all_gather_records(..., rows=shard_rows(N, W))` -> [N, record] in global item order
    (drops the padding rows of the short shards)."""
    world = allrec.shape[0]
    parts = [allrec[r, :len(shard_indices(num_items, r, world))] for r in range(world)]
    return torch.cat(parts, 0) if parts else allrec.reshape(0, allrec.shape[-1])


def all_gather_records(local: torch.Tensor, group=None, rows: int = None) -> torch.Tensor:
    """[n_local, record] on every rank -> [world, rows, record]; one collective.  `all_gather_into_tensor` needs the
    same row count on every rank: with `rows` (= `shard_rows(N, W)`) a short shard is padded with all-zero records
    (count 0) and an oversized one raises.  Without `rows` every rank must pass the same n_local (N % W == 0): the ranks
    first exchange their row counts (one more tiny collective and a host read-back - pass `rows` on a hot path) and ALL of
    them raise if the counts differ, instead of hanging or mis-viewing inside the payload collective."""
    import torch.distributed as dist
    oversize = None
    if rows is not None:
        if local.shape[0] > rows:
            # raising HERE would leave the other ranks waiting inside the collective: take part with a truncated, poisoned
            # shard (count -1 in every record: unpack_* raise on it on every rank) and raise after the collective has run
            oversize = local.shape[0]
            local = local[:rows].clone()
            local[:, 0] = -1.0
        if local.shape[0] < rows:
            pad = torch.zeros((rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], 0)
    if not (dist.is_available() and dist.is_initialized()):
        if oversize is not None:
            raise ValueError(f"all_gather_records: {oversize} local records > rows={rows}")
        return local.unsqueeze(0)
    # (a ONE-rank process group still runs the collective: that is how RCCL's all_gather is put through this code path on a
    #  1-GPU box - bench.py GLASS_BENCH_RCCL_WORLD1, tests/test_gpu_h_sharded.py)
    world = dist.get_world_size(group)
    if rows is None:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        counts = torch.empty((world,), dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(counts, mine, group=group)
        counts = counts.tolist()
        if len(set(counts)) != 1:
            raise ValueError(f"all_gather_records: ranks hold different record counts {counts}; pass rows=shard_rows(N, W)")
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)      # rank-major concatenation
    if oversize is not None:
        raise ValueError(f"all_gather_records: {oversize} local records > rows={rows} (the other ranks received this shard's "
                         f"records with count -1 and raise when they unpack them)")
    return out.view((world,) + tuple(local.shape))


def _record_count(rec: torch.Tensor, i: int) -> int:
    k = int(rec[i, 0].item())
    if k < 0:
        raise ValueError(f"record {i} is poisoned (count {k}): the rank that owns it passed more records than `rows` to "
                         f"all_gather_records")
    return k


# --------------------------------------------------------------------------- launching the ranks
# The reference starts its ranks itself: tools/eval_glass.py:199-206 `launch(main, num_gpus, num_machines=1, machine_rank=0,
# dist_url=..., args=(args,))` (detectron2.engine.launch: one process per GPU on this node, NCCL process group, then main()).
# Same contract here without the pickled closure: re-exec a command once per rank with the torch.distributed.run
# environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), so a script works identically under `torchrun` and
# when it launches itself.

def free_port(addr: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world_size: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """environment of local rank `rank` of a one-node job (what `python -m torch.distributed.run --nnodes=1` exports)"""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size), "LOCAL_WORLD_SIZE": str(world_size),
                "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL across processes needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")                      # torchrun's default for nproc > 1: the host side is launch glue
    return env


def _die_with_parent() -> None:
    """child side (between fork and exec): SIGKILL this rank when the launcher dies, however it dies (Linux PR_SET_PDEATHSIG) -
    the ranks sit in their own sessions, so a signal aimed at the launcher's process group would not reach them and they would
    keep their GPUs"""
    try:
        import ctypes
        import signal
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)       # 1 = PR_SET_PDEATHSIG
    except Exception:                                                                          # noqa: BLE001 - best effort
        pass


class _Terminated(BaseException):
    pass


def launch_local_ranks(argv: Sequence[str], world_size: int, port: Optional[int] = None, base_env: Optional[Dict[str, str]] = None,
                       poll_s: float = 0.05, grace_s: float = 5.0) -> int:
    """Start `argv` once per rank (own session each, stdio inherited), wait for all of them; the first non-zero exit code
    terminates the remaining ranks (SIGTERM to exactly the process groups started here, SIGKILL after `grace_s`) and is
    returned - a rank that died must not leave the others waiting in a collective.  0 = every rank exited 0.
    The ranks never outlive the launcher: SIGTERM / SIGINT / SIGHUP to the launcher stop them the same way (and return
    128 + signal), and a launcher that is killed outright takes them along (`_die_with_parent`).
    Call it from the MAIN thread: the signal handlers can only be installed there, and Linux delivers PR_SET_PDEATHSIG when
    the THREAD that forked the rank exits - from a short-lived worker thread the ranks would be killed when that thread
    returns, so off the main thread the ranks are started without it (and then only the explicit stop paths apply)."""
    import signal
    import threading
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    port = free_port() if port is None else int(port)
    on_main = threading.current_thread() is threading.main_thread()
    procs = [subprocess.Popen(list(argv), env=rank_env(r, world_size, port, base_env), start_new_session=True,
                              preexec_fn=_die_with_parent if on_main else None)
             for r in range(world_size)]
    rc = 0
    old_handlers = {}
    if on_main:
        def on_signal(signum, _frame):
            raise _Terminated(signum)
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            try:
                old_handlers[sg] = signal.signal(sg, on_signal)
            except (OSError, ValueError):
                pass
    try:
        live = set(range(world_size))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 128 - code        # a signal -N reads as 128 + N, like a shell
                        print(f"[launch_local_ranks] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        break
            if live and rc == 0:
                time.sleep(poll_s)
    except _Terminated as e:
        rc = 128 + int(e.args[0])
        print(f"[launch_local_ranks] signal {int(e.args[0])}: stopping the ranks", file=sys.stderr)
    finally:
        def stop_ranks():
            for sg in old_handlers:
                signal.signal(sg, signal.SIG_IGN)                    # no second interruption while the ranks are being stopped
            for sig, wait in ((signal.SIGTERM, grace_s), (signal.SIGKILL, grace_s)):
                left = [p for p in procs if p.poll() is None]
                if not left:
                    break
                for p in left:
                    try:
                        os.killpg(p.pid, sig)                        # start_new_session: pgid == pid of the rank we started
                    except ProcessLookupError:
                        pass
                t_end = time.time() + wait
                while time.time() < t_end and any(p.poll() is None for p in left):
                    time.sleep(poll_s)
        # a signal that lands after the poll loop but before SIG_IGN is installed raises _Terminated INSIDE this block: the
        # kill loop must still run (it was skipped, ranks survived a doubly-signalled launcher) - retry until it has
        while True:
            try:
                stop_ranks()
                break
            except _Terminated as e:
                rc = rc or 128 + int(e.args[0])
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


# ---------------------------------------------------------------------------------------------- first-contact hardening
# The N-GPU path (RCCL over xGMI) cannot be exercised from the build container (1-GPU leases): the first run is the
# driver's.  Everything below makes that run fail FAST and LEGIBLY instead of hanging: a rendezvous timeout, a preflight
# report before any model is built, and each rank pinned to the CPUs of its GPU's NUMA node.

RENDEZVOUS_TIMEOUT_S = 120


def init_process_group(backend: str, device: Optional[torch.device] = None, timeout_s: Optional[float] = None):
    """torch.distributed.init_process_group with a rendezvous / collective timeout of `timeout_s` (default 120 s, or
    $GLASS_DIST_TIMEOUT_S) instead of torch's 10-30 minutes: a rank that never shows up makes every other rank raise within
    the timeout and exit non-zero (bench.py's launcher then stops the rest).  Returns the timeout used (seconds)."""
    import datetime
    import torch.distributed as dist
    t = float(os.environ.get("GLASS_DIST_TIMEOUT_S", RENDEZVOUS_TIMEOUT_S) if timeout_s is None else timeout_s)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"timeout": datetime.timedelta(seconds=t)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return t


def gpu_numa_cpus(dev_index: int) -> Tuple[Optional[int], List[int], str]:
    """(numa node, its CPUs, PCI address) of CUDA/HIP device `dev_index` from sysfs; node None when the platform does not say
    (single-node hosts report -1) or the device cannot be located."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        # a torch build without the pci_* attributes must NOT fabricate 0000:00:00.0 - that address exists in sysfs (the host
        # bridge, usually NUMA node 0) and every rank would be pinned to node 0 while the report says "pinned" 
        dom, bus, dv = (getattr(pr, n, None) for n in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if dom is None or bus is None or dv is None:
            return None, [], "?"
        bdf = f"{int(dom):04x}:{int(bus):02x}:{int(dv):02x}.0"
    except Exception:                                            # noqa: BLE001 - no device: nothing to pin to
        return None, [], "?"
    return _numa_cpus_of_pci(bdf) + (bdf,)


def _numa_cpus_of_pci(bdf: str, sysfs: str = "/sys") -> Tuple[Optional[int], List[int]]:
    try:
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read().strip())
    except (OSError, ValueError):
        return None, []
    if node < 0:
        return None, []
    try:
        return node, parse_cpulist(open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")).read())
    except OSError:
        return node, []


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the sysfs cpulist format)"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def pin_to_gpu_numa_node(dev_index: int) -> Dict[str, object]:
    """Restrict this process to the CPUs of its GPU's NUMA node (intersection with the CPUs it may already use; nothing
    happens when the node is unknown or the intersection is empty - a pin must never take CPUs away entirely).  Launch glue
    and pinned-memory copies then stay on the socket the GPU hangs off.  Returns a report for the preflight line."""
    node, cpus, bdf = gpu_numa_cpus(dev_index)
    rep: Dict[str, object] = {"device": dev_index, "pci": bdf, "numa_node": node, "pinned": False}
    try:
        have = os.sched_getaffinity(0)
    except AttributeError:                                       # pragma: no cover (non-linux)
        return rep
    rep["cpus_before"] = len(have)
    want = have & set(cpus)
    if node is not None and want and want != have:
        try:
            os.sched_setaffinity(0, want)
            rep["pinned"] = True
        except OSError as e:
            rep["error"] = str(e)
    rep["cpus_after"] = len(os.sched_getaffinity(0))
    return rep


def preflight_report(world_size: int, backend: str) -> Dict[str, object]:
    """what a failed multi-GPU run needs on its first line: visible devices, RCCL version, the IPC mode the host driver needs,
    the rendezvous address - gathered WITHOUT touching a device context or a process group"""
    rep: Dict[str, object] = {"world_size": world_size, "backend": backend, "cuda_available": bool(torch.cuda.is_available()),
                              "devices_visible": torch.cuda.device_count() if torch.cuda.is_available() else 0,
                              "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                              "MASTER_ADDR": os.environ.get("MASTER_ADDR"), "MASTER_PORT": os.environ.get("MASTER_PORT"),
                              "timeout_s": float(os.environ.get("GLASS_DIST_TIMEOUT_S", RENDEZVOUS_TIMEOUT_S)), "torch": torch.__version__,
                              "hip": getattr(torch.version, "hip", None)}
    try:
        rep["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                       # noqa: BLE001 - a report field, never fatal
        rep["rccl_version"] = f"unavailable ({type(e).__name__})"
    return rep
