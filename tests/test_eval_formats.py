"""Evaluation wire formats against records / lines produced by the reference's own functions
(oracle/make_golden.py --eval -> tests/golden/eval_formats.{json,npz}) and known answers for the pieces whose
third-party dependencies (shapely, python-Levenshtein) are restated."""
import io
import json
import os
import zipfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _setup():
    from glass_amd.config import get_glass_cfg
    from glass_amd.modeling.recognition.text_encoder import TextEncoder
    cfg = get_glass_cfg()
    cfg.MODEL.ROI_RECOGNIZER_HEAD.NAME = "RecognizerRCNNHeadV3"
    cfg.MODEL.ROI_RECOGNIZER_HEAD.MAX_WORD_LENGTH = 25
    return TextEncoder(cfg), json.load(open(os.path.join(GOLD, "eval_formats.json"))), np.load(os.path.join(GOLD, "eval_formats.npz"))


def _instances(img, tp):
    from glass_amd.structures.core import Instances, RotatedBoxes
    inst = Instances((480, 640))
    b = torch.tensor(img["boxes"])
    inst.pred_boxes, inst.pred_rboxes = RotatedBoxes(b.clone()), RotatedBoxes(b.clone())
    inst.scores = torch.tensor(img["scores"])
    inst.pred_classes = torch.zeros(len(b), dtype=torch.int64)
    inst.pred_text_prob = torch.from_numpy(tp)
    return inst


def test_records_equal_reference_instances_to_coco_json():
    from glass_amd.evaluation import instances_to_coco_json
    enc, gold, arr = _setup()
    for fname, img in gold["images"].items():
        got = instances_to_coco_json(_instances(img, arr[fname + ":text_prob"]), fname, enc, True)
        want = img["records"]
        assert len(got) == len(want)
        cp = arr[fname + ":character_probs"]
        for i, (g, w) in enumerate(zip(got, want)):
            assert g["rec"] == w["rec"] and g["image_id"] == w["image_id"] and g["category_id"] == 1
            for k in ("polys", "boxes", "rboxes"):
                np.testing.assert_allclose(np.asarray(g[k]), np.asarray(w[k]), rtol=0, atol=1e-9)
            assert abs(g["score_text"] - w["score_text"]) < 1e-12 and abs(g["score_detection"] - w["score_detection"]) < 1e-12
            np.testing.assert_allclose(np.asarray(g["character_probs"]), cp[i], rtol=0, atol=0)
    from glass_amd.structures.core import Instances
    empty = Instances((4, 4))
    empty.scores = torch.zeros((0,))
    assert instances_to_coco_json(empty, "x", enc, True) == []


def test_eval_lines_equal_reference_to_eval_format_and_match_transcript():
    from glass_amd.evaluation import TextResultWriter, match_transcript
    enc, gold, arr = _setup()
    for mode, ws in (("e2e", False), ("word_spotting", True)):
        w = TextResultWriter(enc, dataset="icdar15", word_spotting=ws)
        w.process([{"file_name": f} for f in gold["images"]],
                  [{"instances": _instances(img, arr[f + ":text_prob"])} for f, img in gold["images"].items()])
        recs = w.coco_results()
        assert [(r["image_id"], r["rec"]) for r in recs] == [(r["image_id"], r["rec"]) for r in gold["flat_records"]]
        files = w.to_eval_format(recs, 0.5, 0.4)
        assert files == gold["eval_files"][mode]
    for t, (plain, spotting) in gold["match_transcript"].items():
        assert match_transcript(t, False) == plain and match_transcript(t, True) == spotting


def test_polygon_normalisation_lexicon_and_zip_known_answers():
    from glass_amd.evaluation import TextResultWriter, find_match_word, levenshtein, normalize_detection_line
    # shapely semantics restated: positive shoelace area (is_ccw) -> reversed; bow-tie and degenerate rings dropped
    assert normalize_detection_line("0,0,10,0,10,5,0,5,####ab") == "0,5,10,5,10,0,0,0,####ab"
    assert normalize_detection_line("0,0,0,5,10,5,10,0,####ab") == "0,0,0,5,10,5,10,0,####ab"
    assert normalize_detection_line("0,0,10,5,10,0,0,5,####x") is None            # self-intersecting
    assert normalize_detection_line("0,0,5,5,10,10,####x") is None                 # zero area
    assert normalize_detection_line("1,1,5,1,####x") is None                       # two points
    assert normalize_detection_line("0,0,4,0,4,4,####a,b") == "4,4,4,0,0,0,####a,b"  # transcription with a comma survives
    assert [levenshtein(a, b) for a, b in (("", ""), ("abc", ""), ("kitten", "sitting"), ("flaw", "lawn"), ("A", "a"))] == [0, 3, 3, 2, 1]
    lex = ["apple", "Maple", "ample"]
    pairs = {w.upper(): w for w in lex}
    assert find_match_word("appel", lex, pairs) == ("apple", 2)
    assert find_match_word("MAPLE", lex, pairs) == ("Maple", 0)
    assert find_match_word("zzz", [], {}) == ("", 100)
    enc, gold, arr = _setup()
    w = TextResultWriter(enc, dataset="totaltext", lexicon=lex, pairs=pairs, lexicon_type=2, edit_distance_thr=1.5)
    recs = [{"image_id": 7, "polys": [[0, 0], [9.7, 0], [9.7, 4.2], [0, 4.2]], "rec": "aple", "score_text": 0.91, "score_detection": 0.8},
            {"image_id": 7, "polys": [[0, 0], [9, 0], [9, 4], [0, 4]], "rec": "qqqqq", "score_text": 0.9, "score_detection": 0.8},   # no lexicon match
            {"image_id": 8, "polys": [[0, 0], [9, 0], [9, 4], [0, 4]], "rec": "maple", "score_text": 0.0005, "score_detection": 0.9}]  # score_text <= 0.001
    files = w.to_eval_format(recs, 0.5, 0.0)
    assert files == {"0000007.txt": ["0,0,9,0,9,4,0,4,####apple"]}
    z = zipfile.ZipFile(io.BytesIO(w.det_zip(files)))
    assert z.namelist() == ["0000007.txt"] and z.read("0000007.txt").decode() == "0,4,9,4,9,0,0,0,####apple\n"


def test_masks_to_polygons_ring_tracer_known_answers():
    """stand-in for rasterio + shapely (absent): exterior ring of the largest 4-connected region on the pixel-corner
    lattice.  Checked on hand-made shapes and by the shoelace area (= pixel count for hole-free regions)."""
    from glass_amd.evaluation import masks_to_polygons

    def area(ring):
        return abs(sum(ring[i][0] * ring[i + 1][1] - ring[i + 1][0] * ring[i][1] for i in range(len(ring) - 1))) / 2

    rect = np.zeros((6, 8), bool); rect[1:4, 2:6] = True
    assert masks_to_polygons([rect]) == [[[2.0, 1.0], [6.0, 1.0], [6.0, 4.0], [2.0, 4.0], [2.0, 1.0]]]
    ell = np.zeros((6, 6), bool); ell[1:5, 1:3] = True; ell[3:5, 3:5] = True
    assert masks_to_polygons([ell]) == [[[1.0, 1.0], [3.0, 1.0], [3.0, 3.0], [5.0, 3.0], [5.0, 5.0], [1.0, 5.0], [1.0, 1.0]]]
    holed = np.ones((5, 5), bool); holed[2, 2] = False                        # exterior ring only
    assert masks_to_polygons([holed]) == [[[0.0, 0.0], [5.0, 0.0], [5.0, 5.0], [0.0, 5.0], [0.0, 0.0]]]
    two = np.zeros((5, 9), bool); two[1:3, 1:3] = True; two[1:4, 5:8] = True    # the larger region wins
    assert area(masks_to_polygons([two])[0]) == 9
    diag = np.zeros((4, 4), bool); diag[0, 0] = diag[1, 1] = diag[1, 2] = diag[2, 1] = True   # diagonal contact only
    assert area(masks_to_polygons([diag])[0]) == 3
    assert masks_to_polygons([np.zeros((3, 3), bool)]) == [[]]
    rng = np.random.RandomState(0)
    from scipy import ndimage
    for _ in range(20):                                                        # random blobs: area = pixels + holes
        m = ndimage.binary_opening(rng.rand(24, 24) > 0.45)
        lab, n = ndimage.label(m)
        if n == 0:
            continue
        k = int(np.argmax(np.bincount(lab.ravel())[1:])) + 1
        reg = lab == k
        ring = masks_to_polygons([m])[0]
        assert ring[0] == ring[-1] and len(ring) >= 5
        filled = ndimage.binary_fill_holes(reg, structure=np.ones((3, 3)))    # holes = background not 8-connected outside
        assert area(ring) == int(filled.sum())
