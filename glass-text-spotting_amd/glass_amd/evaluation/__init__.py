from .text_evaluator import (TextResultWriter, boxes_to_polygons, find_match_word, instances_to_coco_json,  # noqa: F401
                             levenshtein, masks_to_polygons, match_transcript, normalize_detection_line, rotated_boxes_to_polygons)
