"""GPU: BASELINE configs[3] - the ICDAR15 cfg, a 64-image global batch sharded over 8 ranks, results exchanged by ONE all_gather
of per-image word records (the only BASELINE config no -m gpu test ran).

A 1-GPU box cannot host eight RCCL ranks, so the sharded path is executed the two ways it can be:
  (i)  eight consecutive "ranks" in this process, each `shard_indices` -> `model.inference` -> word post-processor ->
       `pack_words` -> `all_gather_records` on a ONE-rank `nccl` process group (RCCL's communicator and its
       all_gather_into_tensor really run, on device tensors), stacked rank-major and re-ordered by `gathered_to_global`;
  (ii) two self-launched ranks (`launch_local_ranks`, the launcher bench.py uses) sharing cuda:0, 32 images each, `gloo`
       collective on host copies of the device records.
Both must give the same [64, record] tensor bit for bit (the steps see the same 8-image batches), in global image order, and
sampled images' records must equal what the CPU oracle (`oracle.glass_cpu.glass_inference`) + the host post-processor
(`PostProcessorAcademic.host_call`, pinned on the reference's goldens) produce for that image ALONE.
A 61-image batch covers the short shards: ceil(61/8) = 8 rows per rank, the ranks that hold 7 images pad with a count-0 record.

Reference: tools/eval_glass.py:200-207, glass/data/build.py:99, glass/evaluation/text_evaluator.py:246-249."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sharded_rank as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, WORLD = 64, 8


@pytest.fixture(scope="module")
def job():
    return S.build()                     # cfg, state dict, model, post-processor


@pytest.fixture(scope="module")
def rccl_world1():
    """a one-rank process group on the RCCL backend (what bench.py's GLASS_BENCH_RCCL_WORLD1 builds)"""
    import torch.distributed as dist
    from glass_amd.distributed import free_port, init_process_group
    assert not dist.is_initialized()
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "RANK": "0", "WORLD_SIZE": "1"})
    torch.cuda.set_device(0)
    init_process_group("nccl", device=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _consecutive_ranks(job, num_items, world, group_dist):
    """(i): every rank's step in turn; each goes through the collective of the one-rank RCCL group"""
    cfg, _, model, post = job
    per_rank = []
    for r in range(world):
        g = S.rank_step(cfg, model, post, num_items, r, world)
        assert g.is_cuda and g.shape[0] == 1, "the one-rank group's all_gather returns [1, rows, record] on the device"
        per_rank.append(g[0])
    return torch.stack(per_rank, 0)       # [world, rows, record]: what a world-8 all_gather_into_tensor lays out rank-major


@pytest.fixture(scope="module")
def global64(job, rccl_world1):
    from glass_amd.distributed import gathered_to_global, shard_rows
    allrec = _consecutive_ranks(job, N, WORLD, rccl_world1)
    assert rccl_world1.get_backend() == "nccl"
    assert tuple(allrec.shape[:2]) == (WORLD, shard_rows(N, WORLD)) == (8, 8)
    return gathered_to_global(allrec, N).cpu()


def _oracle_words(job, g):
    """image g ALONE: CPU oracle (recognition of its 32 injected boxes) -> meta-arch postprocess -> host post-processor"""
    from glass_amd.structures.core import Instances, RotatedBoxes
    from glass_amd.utils.synth import make_boxes, make_image
    from oracle import glass_cpu as O
    cfg, sd, _, post = job
    img = make_image(g, S.SIDE, S.SIDE).permute(2, 0, 1).float().contiguous()
    boxes = make_boxes(g, S.ROIS, S.SIDE, S.SIDE)
    ref = O.glass_inference(sd, [img], cfg, injected_boxes=[boxes])[0]
    det = {"pred_boxes": boxes, "scores": torch.ones(len(boxes)), "pred_classes": torch.zeros(len(boxes), dtype=torch.int64),
           "orientations": torch.zeros((len(boxes), 2)), "pred_text_prob": ref["pred_text_prob"]}
    det = O.meta_postprocess(det, (S.SIDE, S.SIDE), (S.SIDE, S.SIDE), cfg.POST_PROCESSING.MIN_BOX_DIMENSION)
    inst = Instances((S.SIDE, S.SIDE))
    inst.pred_boxes = RotatedBoxes(det["pred_boxes"].cuda())
    inst.scores = det["scores"].cuda()
    inst.pred_classes = det["pred_classes"].cuda()
    inst.orientations = det["orientations"].cuda()
    inst.pred_text_prob = det["pred_text_prob"].cuda()
    return post.host_call(inst)


def _compare_with_oracle(job, rec, g):
    from glass_amd.distributed import unpack_words
    from glass_amd.postprocess.post_processor_academic import get_instances_text, strip_special
    cfg, _, _, post = job
    max_det, steps_txt = S.record_dims(cfg)
    got = unpack_words(rec[g:g + 1], max_det, steps_txt, post.text_encoder.character)[0]
    ref = _oracle_words(job, g)
    n = len(ref)
    assert len(got["boxes"]) == n > 0, f"image {g}: {len(got['boxes'])} words vs oracle {n}"
    rb = ref.pred_boxes.tensor.cpu().numpy().astype(np.float64)
    gb = got["boxes"].numpy().astype(np.float64)
    db = np.abs(gb - rb)
    db[:, 4] = np.abs((gb[:, 4] - rb[:, 4] + 180.0) % 360.0 - 180.0)
    dpoly = float(np.abs(got["polygons"].numpy() - ref.pred_polygons.cpu().numpy()).max())
    assert db.max() < 2e-3 and dpoly < 2e-3, (g, db.max(), dpoly)
    np.testing.assert_allclose(got["scores"].numpy(), ref.scores.cpu().numpy(), atol=1e-6)
    # texts: the greedy decoder feeds its arg-max back, so a word whose oracle probabilities hold a near-tie (two best classes
    # closer than 1e-4 at a live step) may legitimately read differently from there on - compared where there is none
    tp = ref.pred_text_prob.cpu().numpy()
    srt = np.sort(tp, axis=-1)
    tied = (((srt[..., -1] - srt[..., -2]) < 1e-4) & (tp.sum(-1) > 0)).any(1)
    texts, tscores, _ = get_instances_text(ref.pred_text_prob, post.text_encoder)
    same = 0
    for j in range(n):
        if tied[j]:
            continue
        assert strip_special(got["texts"][j]) == texts[j], (g, j, got["texts"][j], texts[j])
        assert abs(float(got["text_scores"][j]) - float(tscores[j])) < 1e-3
        same += 1
    print(f"[parity] configs[3] image {g}: {n} words, max |dbox| {db.max():.2e}, max |dpolygon| {dpoly:.2e} px, "
          f"{same} texts identical, {int(tied.sum())} skipped at a near-tie")
    assert same >= 0.9 * n
    return n


def test_config3_sharded_64_images_on_a_one_rank_rccl_group_vs_oracle(job, global64):
    """(i) + the oracle: global order, one record per image, 8 sampled images (both ends of several shards) against the
    oracle of that image alone"""
    cfg = job[0]
    max_det, steps_txt = S.record_dims(cfg)
    from glass_amd.distributed import words_record_size
    assert tuple(global64.shape) == (N, words_record_size(max_det, steps_txt))
    counts = global64[:, 0]
    assert bool((counts > 0).all()) and bool((counts <= max_det).all()), "every image carries words (TEXT_THRESHOLD 0)"
    # global order: no two images share a record (distinct seeds -> distinct boxes), and the sampled ones are the oracle's
    assert len({tuple(np.round(global64[g, 1:6].numpy(), 3)) for g in range(N)}) == N
    total = sum(_compare_with_oracle(job, global64, g) for g in (0, 7, 8, 21, 31, 32, 45, 63))
    assert total > 8 * 10


def test_config3_two_self_launched_ranks_sharing_the_gpu_give_the_same_global_records(job, global64, tmp_path):
    """(ii): two processes started by `launch_local_ranks` (own sessions, torch.distributed.run environment), 32 images each,
    gloo all_gather of the records -> bit-identical to (i)"""
    from glass_amd.distributed import launch_local_ranks
    out = str(tmp_path / "global.npy")
    torch.cuda.synchronize()
    rc = launch_local_ranks([sys.executable, os.path.join(ROOT, "tests", "sharded_rank.py"), str(N), out], 2)
    assert rc == 0, f"a rank failed (exit {rc})"
    got = np.load(out)
    assert got.shape == tuple(global64.shape)
    assert np.array_equal(got, global64.numpy()), "two gloo ranks and eight consecutive RCCL-world-1 ranks disagree"


def test_config3_short_shards_are_padded_with_count_zero_records(job, global64, rccl_world1):
    """61 images over 8 ranks: shards of 7 or 8, `rows` = 8; a 7-image rank pads its contribution with one all-zero record
    (count 0), `gathered_to_global` drops the padding and keeps the global order.  (A 7-image step routes some layers
    differently from an 8-image one, so the records equal the 64-image run's to fp32 summation order, not bit for bit.)"""
    from glass_amd.distributed import gathered_to_global, shard_indices, shard_rows
    n = 61
    allrec = _consecutive_ranks(job, n, WORLD, rccl_world1).cpu()
    rows = shard_rows(n, WORLD)
    assert rows == 8 and tuple(allrec.shape[:2]) == (WORLD, rows)
    short = [r for r in range(WORLD) if len(shard_indices(n, r, WORLD)) < rows]
    assert short and len(short) == WORLD * rows - n
    for r in range(WORLD):
        k = len(shard_indices(n, r, WORLD))
        assert bool((allrec[r, :k, 0] > 0).all())
        assert bool((allrec[r, k:] == 0).all()), "padding rows are all-zero records (count 0)"
    glob = gathered_to_global(allrec, n)
    assert glob.shape[0] == n
    cfg = job[0]
    D = S.record_dims(cfg)[0]
    for g in range(n):
        assert glob[g, 0] == global64[g, 0], f"image {g}: word count differs from the 64-image run"
        c = int(glob[g, 0])
        np.testing.assert_allclose(glob[g, 1:1 + 5 * D].view(D, 5)[:c].numpy(), global64[g, 1:1 + 5 * D].view(D, 5)[:c].numpy(), atol=2e-3)
