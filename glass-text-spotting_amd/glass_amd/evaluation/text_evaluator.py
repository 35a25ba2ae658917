"""Evaluation wire formats: what the reference's `TextEvaluator` writes, without its protocol.

Host-side only (strings, JSON, zip): nothing here touches the GPU.  Mirrors, with the reference's names:
  * `instances_to_coco_json`  - one record per recognised word: polys / boxes / rboxes / rec / score_text /
    character_probs / score_detection (glass/evaluation/text_evaluator.py:351-415),
  * `boxes_to_polygons`, `rotated_boxes_to_polygons` (:418-461),
  * `match_transcript` (:299-321), `find_match_word` (lexicon_utils.py:4-49, plain edit distance; the python
    `Levenshtein` package is replaced by `levenshtein` below),
  * `TextResultWriter.to_eval_format` / `sort_detection` (:96-239): the RRC "x1,y1,...,xn,yn,####text" files, one
    per image, thresholded, clockwise, zipped as det.zip.
`masks_to_polygons` (:464-492) is a pixel-edge ring tracer standing in for the rasterio + shapely polygoniser the
reference uses (absent here); pass it (or your own) through `masks_to_polygons=`, otherwise the rotated box
polygon is used.  The scoring itself (`text_eval_script`, the
official RRC script) and dataset catalogues are out of scope.
"""
from __future__ import annotations

import io
import json
import os
import re
import zipfile
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..postprocess.post_processor_academic import get_instances_text

_SPECIAL = str("'!?.:,*+\"()·[]/\\#$%;<=>@^_`{|}~")


def boxes_to_polygons(boxes: np.ndarray) -> np.ndarray:
    """XYXY boxes -> 4 corners (x0y0, x1y0, x1y1, x0y1); the reference applies it to whatever `pred_boxes` holds,
    i.e. to the first four columns (cx, cy, w, h) of rotated boxes as well (text_evaluator.py:375-376)."""
    n = len(boxes)
    if n == 0:
        return np.array([]).reshape((0, 4, 2))
    p = np.zeros((n, 4, 2))
    p[:, 0, 0], p[:, 0, 1] = boxes[:, 0], boxes[:, 1]
    p[:, 1, 0], p[:, 1, 1] = boxes[:, 2], boxes[:, 1]
    p[:, 2, 0], p[:, 2, 1] = boxes[:, 2], boxes[:, 3]
    p[:, 3, 0], p[:, 3, 1] = boxes[:, 0], boxes[:, 3]
    return p


def rotated_boxes_to_polygons(boxes: np.ndarray) -> np.ndarray:
    n = len(boxes)
    if n == 0:
        return np.array([]).reshape((0, 4, 2))
    assert boxes.shape[-1] == 5, "The last dimension of input shape must be 5 for XYWHA format"
    cx, cy, w, h, a = (boxes[:, i] for i in range(5))
    t = np.deg2rad(-a)
    s, c = np.sin(t), np.cos(t)
    p = np.zeros((n, 4, 2))
    p[:, 0, 0] = cx + (h * s - w * c) / 2
    p[:, 1, 0] = cx + (h * s + w * c) / 2
    p[:, 2, 0] = cx - (h * s - w * c) / 2
    p[:, 3, 0] = cx - (h * s + w * c) / 2
    p[:, 0, 1] = cy - (h * c + w * s) / 2
    p[:, 1, 1] = cy - (h * c - w * s) / 2
    p[:, 2, 1] = cy + (h * c + w * s) / 2
    p[:, 3, 1] = cy + (h * c - w * s) / 2
    return p


def masks_to_polygons(masks: np.ndarray) -> list:
    """Largest 4-connected region of each boolean mask as its exterior ring along the pixel edges: what the reference
    gets from `rasterio.features.shapes` + shapely (`masks_to_polygons`, text_evaluator.py:464-492: keep the polygon of
    largest area, return `exterior.coords` as [[x, y], ...], closed).  rasterio / GDAL are absent here, so the ring is
    traced directly [third-party recall: GDAL polygonises 4-connected regions on the pixel-corner lattice]: vertices
    are pixel corners (x = column, y = row), only corners where the boundary turns are emitted, the ring is closed
    (first point repeated).  Vertex start / winding are this tracer's own (clockwise in image coordinates);
    downstream `sort_detection` re-orients rings and the protocol only uses the geometry.  Area ties keep the first
    region in scan order."""
    from scipy import ndimage
    out = []
    four = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    for mask in masks:
        m = np.asarray(mask).astype(bool)
        lab, n = ndimage.label(m, structure=four)
        if n == 0:
            out.append([])
            continue
        sizes = np.bincount(lab.ravel())[1:]
        k = int(np.argmax(sizes)) + 1                       # argmax returns the first maximum
        reg = np.pad(lab == k, 1)                           # padded: reg[r + 1, c + 1] is pixel (r, c)
        rows, cols = np.nonzero(reg)
        r0 = int(rows.min())
        c0 = int(cols[rows == r0].min())
        # boundary walk on the corner lattice with the region on the RIGHT hand side (clockwise in image coordinates),
        # starting at the top-left corner of the top-left-most pixel, heading east.  Directions: 0 E, 1 S, 2 W, 3 N.
        # At a corner (y, x) the four pixels around it are NW = reg[y-1, x-1], NE = reg[y-1, x], SW = reg[y, x-1],
        # SE = reg[y, x] (padded coordinates).
        def px(y, x):
            return bool(reg[y, x])
        y, x, d = r0, c0, 0
        start = (y, x, d)
        ring = [(x - 1, y - 1)]
        while True:
            # advance one edge
            if d == 0:
                x += 1
            elif d == 1:
                y += 1
            elif d == 2:
                x -= 1
            else:
                y -= 1
            nw, ne, sw, se = px(y - 1, x - 1), px(y - 1, x), px(y, x - 1), px(y, x)
            # candidates in priority order: turn right, straight, turn left (region kept on the right; at a diagonal
            # pinch the right turn keeps the walk on the same 4-connected region)
            ahead_right = {0: se, 1: sw, 2: nw, 3: ne}[d]    # pixel ahead on the right-hand side
            ahead_left = {0: ne, 1: se, 2: sw, 3: nw}[d]     # pixel ahead on the left-hand side
            if not ahead_right:
                nd = (d + 1) % 4                              # region ends: turn right around it
            elif ahead_left:
                nd = (d + 3) % 4                              # blocked ahead: turn left
            else:
                nd = d
            if nd != d:
                ring.append((x - 1, y - 1))
            d = nd
            if (y, x, d) == start:
                break
        if ring[-1] != ring[0]:
            ring.append(ring[0])
        out.append([[float(a), float(b)] for a, b in ring])
    return out


def instances_to_coco_json(instances, file_name, text_encoder, onlyRemoveFirstLastCharacter=True,
                           masks_to_polygons: Optional[Callable] = None) -> List[dict]:
    """One dict per word with non-empty text and a polygon of >= 3 points (text_evaluator.py:351-415)."""
    if len(instances) == 0:
        return []
    if instances.has("pred_masks") and masks_to_polygons is not None:
        polygons = masks_to_polygons(instances.pred_masks.cpu().numpy())
    else:
        b = instances.pred_boxes.tensor.cpu().numpy()
        polygons = (boxes_to_polygons(b) if b.shape[1] == 4 else rotated_boxes_to_polygons(b)).tolist()
    rboxes = (rotated_boxes_to_polygons(instances.pred_rboxes.tensor.cpu().numpy()).tolist()
              if instances.has("pred_rboxes") else [[]] * len(polygons))
    boxes = (boxes_to_polygons(instances.pred_boxes.tensor.cpu().numpy()).tolist()
             if instances.has("pred_boxes") else [[]] * len(polygons))
    pred_text, scores_text, text_probs = get_instances_text(instances.pred_text_prob, text_encoder, onlyRemoveFirstLastCharacter)
    scores_detection = instances.scores.tolist()
    results = []
    for poly, rec, st, cp, box, rbox, sd in zip(polygons, pred_text, scores_text, text_probs, boxes, rboxes, scores_detection):
        if len(rec) > 0 and len(poly) >= 3:
            results.append({"image_id": file_name, "category_id": 1, "polys": poly, "boxes": box, "rboxes": rbox, "rec": rec,
                            "score_text": np.float64(st).tolist(), "character_probs": np.float64(cp).tolist(),
                            "score_detection": np.float64(sd).tolist()})
    return results


def match_transcript(transcription: str, word_spotting: bool) -> str:
    """text_evaluator.py:299-321."""
    if word_spotting:
        if transcription[len(transcription) - 2:] in ("'s", "'S"):
            transcription = transcription[0:len(transcription) - 2]
        transcription = transcription.strip("-")
        for ch in _SPECIAL:
            transcription = transcription.replace(ch, " ")
        return transcription.strip()
    if len(transcription) > 0 and _SPECIAL.find(transcription[0]) > -1:
        transcription = transcription[1:]
    if len(transcription) > 0 and _SPECIAL.find(transcription[-1]) > -1:
        transcription = transcription[:-1]
    return transcription


def levenshtein(a: str, b: str) -> int:
    """plain edit distance (insert / delete / substitute, unit costs) = `Levenshtein.distance`."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def find_match_word(rec_str: str, lexicon: Sequence[str], pairs: Dict[str, str]) -> Tuple[str, int]:
    """Closest lexicon word by edit distance on upper-cased strings, first minimum wins
    (lexicon_utils.py:4-28, the un-weighted branch)."""
    dist_min, match_word, match_dist = 100, "", 100
    rec = rec_str.upper()
    for word in lexicon:
        word = word.upper()
        d = levenshtein(rec, word)
        if d < dist_min:
            dist_min, match_word, match_dist = d, pairs[word], d
    return match_word, match_dist


def _segments_cross(p, q, r, s) -> bool:
    def orient(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
    d1, d2, d3, d4 = orient(r, s, p), orient(r, s, q), orient(p, q, r), orient(p, q, s)
    return ((d1 > 0) != (d2 > 0)) and ((d3 > 0) != (d4 > 0)) and d1 != 0 and d2 != 0 and d3 != 0 and d4 != 0


def normalize_detection_line(line: str) -> Optional[str]:
    """One line of `sort_detection` (text_evaluator.py:112-137): drop invalid polygons (fewer than 3 points, zero
    area or self-intersecting - shapely's `is_valid` for a simple ring), make the ring clockwise in image
    coordinates (reversed when `LinearRing.is_ccw`, i.e. positive shoelace area), keep the transcription."""
    ptr = line.strip().split(",####")
    rec = ptr[1]
    cors = ptr[0].split(",")
    assert len(cors) % 2 == 0, "cors invalid."
    pts = [(int(cors[j]), int(cors[j + 1])) for j in range(0, len(cors), 2)]
    n = len(pts)
    if n < 3:
        return None
    area2 = sum(pts[i][0] * pts[(i + 1) % n][1] - pts[(i + 1) % n][0] * pts[i][1] for i in range(n))
    if area2 == 0:
        return None
    for i in range(n):
        for j in range(i + 2, n):
            if i == 0 and j == n - 1:
                continue
            if _segments_cross(pts[i], pts[(i + 1) % n], pts[j], pts[(j + 1) % n]):
                return None
    if area2 > 0:
        pts.reverse()
    return ",".join(f"{int(x)},{int(y)}" for x, y in pts) + ",####" + rec


class TextResultWriter:
    """`TextEvaluator.process` / `to_eval_format` / `sort_detection` without files in the way: collect per-image
    records, then emit {file name -> lines} or a det.zip byte string.  `dataset` picks the reference's file naming
    ('totaltext' / 'textocr': %07d.txt, 'icdar*': %d.txt) and image-id base (totaltext 0, others 1)."""

    def __init__(self, text_encoder, dataset: str = "icdar15", word_spotting: bool = False,
                 onlyRemoveFirstLastCharacter: bool = True, lexicon: Optional[Sequence[str]] = None,
                 pairs: Optional[Dict[str, str]] = None, lexicon_type: Optional[int] = None, edit_distance_thr: float = 1.5,
                 masks_to_polygons: Optional[Callable] = None):
        self.text_encoder, self.dataset, self.word_spotting = text_encoder, dataset, word_spotting
        self.only_first_last = onlyRemoveFirstLastCharacter
        self.lexicon, self.pairs, self.lexicon_type, self.edit_distance_thr = lexicon, pairs, lexicon_type, edit_distance_thr
        self.masks_to_polygons = masks_to_polygons
        self._predictions: List[dict] = []

    def reset(self) -> None:
        self._predictions = []

    def process(self, inputs: Sequence[dict], outputs: Sequence[dict]) -> None:
        for inp, out in zip(inputs, outputs):
            inst = out["instances"]
            self._predictions.append({"file_name": inp["file_name"],
                                      "instances": instances_to_coco_json(inst, inp["file_name"], self.text_encoder,
                                                                          self.only_first_last, self.masks_to_polygons)})

    def coco_results(self) -> List[dict]:
        """`evaluate()` up to text_results.json (:262-281): sort images, assign image ids, flatten."""
        preds = list(self._predictions)
        if self.dataset == "totaltext":
            preds = sorted(preds, key=lambda k: k["file_name"])
        elif self.dataset.startswith("icdar"):
            preds = sorted(preds, key=lambda k: float(re.split(r"([-+]?[0-9]*\.]*)", k["file_name"])[1]))
        out = []
        for i, pred in enumerate(preds):
            image_id = i if self.dataset == "totaltext" else i + 1
            for x in pred["instances"]:
                x = dict(x)
                x["image_id"] = image_id
                out.append(x)
        return out

    def to_eval_format(self, records: Sequence[dict], text_cf_th: float = 0.5, detection_cf_th: float = 0.0) -> Dict[str, List[str]]:
        """records (coco_results) -> {"<id>.txt": ["x1,y1,...,####text", ...]} (:157-239): words with
        score_text <= 0.001 are dropped, non-ASCII characters removed, optional lexicon replacement, transcript
        normalisation for lexicon / word-spotting runs, scores rounded to 3 digits BEFORE thresholding."""
        files: Dict[str, List[str]] = {}
        for d in records:
            if not d["score_text"] > 0.001:
                continue
            cors = ",".join(f"{int(p[0])},{int(p[1])}" for p in d.get("polys", []))
            ass = "".join(c for c in d["rec"] if ord(c) < 128)
            if self.lexicon:
                if self.lexicon_type == 3 and self.dataset.startswith("icdar"):
                    lex, pairs = self.lexicon[d["image_id"]], self.pairs[d["image_id"]]
                else:
                    lex, pairs = self.lexicon, self.pairs
                word, dist = find_match_word(ass, lex, pairs)
                if dist < self.edit_distance_thr or self.lexicon_type == 1:
                    ass = word
                else:
                    continue
            if self.lexicon or self.word_spotting:
                ass = match_transcript(ass, self.word_spotting)
            st, sd = round(d["score_text"], 3), round(d["score_detection"], 3)
            if self.dataset in ("totaltext", "textocr"):
                name = "{:07d}.txt".format(int(d["image_id"]))
            elif self.dataset.startswith("icdar"):
                name = "{}.txt".format(int(d["image_id"]))
            else:
                raise ValueError(self.dataset)
            files.setdefault(name, [])
            if float(str(st)) < text_cf_th or float(str(sd)) < detection_cf_th:
                continue
            files[name].append(cors + ",####" + ass)
        return files

    def det_zip(self, files: Dict[str, List[str]]) -> bytes:
        """`sort_detection` (:96-155): per line validity / orientation normalisation, files zipped as det.zip."""
        buf = io.BytesIO()
        with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED) as z:
            for name in sorted(files):
                lines = [normalize_detection_line(l) for l in files[name]]
                z.writestr(name, "".join(l + "\n" for l in lines if l is not None))
        return buf.getvalue()

    def write(self, output_dir: str, text_cf_th: float = 0.5, detection_cf_th: float = 0.0) -> Tuple[str, str]:
        os.makedirs(output_dir, exist_ok=True)
        records = self.coco_results()
        jpath = os.path.join(output_dir, "text_results.json")
        with open(jpath, "w") as f:
            f.write(json.dumps(records))
        zpath = os.path.join(output_dir, f"{text_cf_th}_{detection_cf_th}", "det.zip")
        os.makedirs(os.path.dirname(zpath), exist_ok=True)
        with open(zpath, "wb") as f:
            f.write(self.det_zip(self.to_eval_format(records, text_cf_th, detection_cf_th)))
        return jpath, zpath
