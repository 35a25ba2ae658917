"""BASELINE configs[3] - "Full GLASS global-to-local fusion, ICDAR15 config, bs=64 sharded over 8 x MI355X via RCCL" - as a
function of (global batch, rank, world): what every rank of the sharded job does, used by tests/test_gpu_h_sharded.py in one
process (eight consecutive "ranks" on a one-rank RCCL group) and, through `python tests/sharded_rank.py`, as a self-launched
rank of a multi-process group (glass_amd.distributed.launch_local_ranks; the ranks of a 1-GPU box share cuda:0).

Reference: tools/eval_glass.py:200-207 (`launch(main, num_gpus, ...)`: one process per GPU), glass/data/build.py:99 (the
inference sampler hands every rank a contiguous shard of the dataset) and glass/evaluation/text_evaluator.py:246-249
(`comm.gather(self._predictions)`).  Here: `shard_indices` -> `model.inference` (8 images per step, 32 injected word boxes each,
the bench's synthetic workload) -> word post-processor -> `pack_words` -> ONE `all_gather_records` -> `gathered_to_global`.
Not a test module itself (no test_ prefix): a helper + a rank entry point.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glass-text-spotting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIDE, ROIS, STEP = 1000, 32, 8            # configs[3]: the ICDAR15 cfg at the metric's image size, 8 images per rank and step
# TEXT_THRESHOLD 0: with random weights every word would fail the shipped 0.25 text-score threshold and the gathered records
# would be empty - the comparison must have words in it
CFG_OPTS = ["MODEL.DEVICE", "cuda:0", "POST_PROCESSING.TEXT_THRESHOLD", 0.0]


def build():
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.postprocess import build_post_processor
    from glass_amd.utils.synth import make_state_dict
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), CFG_OPTS)
    sd = make_state_dict(1234)
    model = glass_amd.build_model(cfg)
    model.load_state_dict(sd)
    return cfg, sd, model, build_post_processor(cfg)


def record_dims(cfg):
    return cfg.TEST.DETECTIONS_PER_IMAGE, cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH + 1


def shard_records(cfg, model, post, indices, side=SIDE, rois=ROIS, step=STEP):
    """word records [len(indices), record] (device) of the global images `indices` (image seed = box seed = global index),
    run as steps of <= `step` images - the per-rank body of the sharded job"""
    import torch
    from glass_amd.distributed import pack_words, words_record_size
    from glass_amd.utils.pipeline import drive
    from glass_amd.utils.synth import make_boxes, make_image
    max_det, steps_txt = record_dims(cfg)
    recs = []
    for k in range(0, len(indices), step):
        g = indices[k:k + step]
        inputs = [{"image": make_image(i, side, side).permute(2, 0, 1).float().contiguous().cuda()} for i in g]
        boxes = [make_boxes(i, rois, side, side).cuda() for i in g]
        det = model.inference(inputs, override_boxes=boxes).batch
        words = drive(post.process_padded_g(det.boxes, det.scores, det.counts_dev, det.text, None, [(side, side)] * len(g),
                                            {"orientations": det.orient}))
        recs.append(pack_words(words.words, max_det, steps_txt))
    if not recs:
        return torch.zeros((0, words_record_size(max_det, steps_txt)), dtype=torch.float32, device="cuda:0")
    return torch.cat(recs, 0)


def rank_step(cfg, model, post, num_items, rank, world, group=None, host_collective=False):
    """rank `rank` of `world`: its shard's records -> the ONE collective -> [world, rows, record]"""
    from glass_amd.distributed import all_gather_records, shard_indices, shard_rows
    local = shard_records(cfg, model, post, shard_indices(num_items, rank, world))
    if host_collective:                    # gloo: the collective runs on host tensors
        local = local.cpu()
    return all_gather_records(local, group=group, rows=shard_rows(num_items, world))


def main() -> int:
    """one self-launched rank (torch.distributed.run environment): gloo group, cuda:0 shared by the ranks of a 1-GPU box;
    rank 0 writes the global records to argv[2] (.npy)"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from glass_amd.distributed import gathered_to_global, init_process_group
    num_items, out_path = int(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    init_process_group("gloo")
    cfg, _, model, post = build()
    allrec = rank_step(cfg, model, post, num_items, rank, world, host_collective=True)
    assert allrec.shape[0] == world
    if rank == 0:
        np.save(out_path, gathered_to_global(allrec, num_items).numpy())
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
