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

from typing import List, Sequence, Tuple

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
    [count | boxes 5D | score D | text score D | polygon 8D | text length D | character index D*T]."""
    N, K = words["scores"].shape
    dev = words["scores"].device
    D = max_det
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
        k = int(rec[i, 0].item())
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
        k = int(rec[i, 0].item())
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
    if rows is not None:
        if local.shape[0] > rows:
            raise ValueError(f"all_gather_records: {local.shape[0]} local records > rows={rows}")
        if local.shape[0] < rows:
            pad = torch.zeros((rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], 0)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.unsqueeze(0)
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
    return out.view((world,) + tuple(local.shape))
