"""Meta-architectures: `GeneralizedRCNN` (d2 flow) and `GlassRCNN` on the HIP pipeline.

Mirrors reference glass/modeling/meta_arch/glass_rcnn.py:13-128 (`GlassRCNN.from_config` :36-55,
`inference` :57-101, `_postprocess` :103-128) and, for `glass_pretrain.yaml` which selects d2's
stock `GeneralizedRCNN` (configs/glass_pretrain.yaml:40), the same flow with d2's
`detector_postprocess` [d2-recall].  Call convention is the reference's:
`model(batched_inputs: list[{'image': Tensor[3,H,W] float 0..255, 'height', 'width'}])
 -> list[{'instances': Instances}]`.

Batches: the reference effectively runs one image per call; a batch here gives,
per image, exactly what the reference gives when run image by image.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ...utils.module import InferenceModule
from ...utils.pipeline import ReadBack, StepOutput, drive

from ...checkpoint import load_checkpoint_file
from ...ops import native as K
from ...postprocess import build_post_processor
from ...postprocess.post_processor_academic import detector_postprocess
from ...structures.core import RotatedBoxes, ImageList, Instances
from ...utils.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
from ..backbone import resnet_fpn as _resnet_fpn  # noqa: F401  (registers the backbone builder)
from ..backbone.resnet_fpn import as_nchw_view
from ..fusion import recognizers_hybrid_head as _hh  # noqa: F401  (registers the ROI head)
from ..proposal_generator import rotated_rpn as _rrpn  # noqa: F401  (registers RotatedRPN)


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(InferenceModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, None)
        shapes = self.backbone.output_shape()
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, shapes)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, shapes)
        self.pixel_mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self.pixel_std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.input_format = cfg.INPUT.FORMAT
        self.conv_precision = str(cfg.MODEL.CONV_PRECISION) if hasattr(cfg.MODEL, "CONV_PRECISION") else "fp32"
        # kernel routing of THIS model (precision from the cfg, the A/B switches from the GLASS_* environment read here, once):
        # stamped on every ConvWeight at load, read by the launches - nothing process-global (SURVEY 8b "Threading")
        self.routing = K.Routing(precision=self.conv_precision)
        self._loaded = False
        # padded device-resident results of the most recent synchronous call.  With several steps in flight
        # (utils.pipeline.run_pipelined) read `.batch` of the step's own return value (pipeline.StepOutput) instead.
        self.last_batch = None

    @property
    def device(self) -> torch.device:
        return self._device

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """d2-keyed state dict (what DetectionCheckpointer hands over) -> folded, re-laid device
        weights.  `roi_heads.mask_head.*` keys are ignored (mask branch off at inference)."""
        dev = self._device
        # every conv / linear weight is re-laid AND packed for this model's conv precision here (checkpoint.conv_weight ->
        # ops.native.prepare_conv_weights): the packed tensors are owned by the layers, so they are freed with the model,
        # and ONE synchronisation at the end makes them visible to whatever streams the steps will run on
        with K.packing_for(self.routing):
            self.backbone.import_weights(state_dict, dev, "backbone.")
            self.proposal_generator.import_weights(state_dict, dev, "proposal_generator.")
            self.roi_heads.import_weights(state_dict, dev, "roi_heads.")
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        self._loaded = True
        return self

    def load_checkpoint(self, path: str):
        return self.load_state_dict(load_checkpoint_file(path))

    # ------------------------------------------------------------------ pipeline
    def preprocess_image(self, batched_inputs: List[Dict], into: Optional[torch.Tensor] = None) -> ImageList:
        """normalise + zero-pad to a multiple of 32 into one NHWC4 batch (glass_rcnn.py:82).  `into`: an existing
        [N,Hp,Wp,4] buffer to fill (the static input of a captured graph)."""
        imgs = [x["image"] for x in batched_inputs]
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in imgs]
        Hp, Wp = ImageList.padded_shape(sizes, self.backbone.size_divisibility)
        batch = into if into is not None else torch.empty((len(imgs), Hp, Wp, 4), dtype=torch.float32, device=self._device)
        assert tuple(batch.shape) == (len(imgs), Hp, Wp, 4)
        for n, im in enumerate(imgs):
            im = im.to(self._device)
            if im.dtype != torch.float32:
                im = im.float()
            K.preprocess_image(im.contiguous(), self.pixel_mean, self.pixel_std, batch, n)
        il = ImageList(as_nchw_view(batch)[:, :3], sizes)
        il.nhwc4 = batch
        return il

    def forward(self, batched_inputs: List[Dict], **kw):
        return self.inference(batched_inputs, **kw)

    def inference(self, batched_inputs: List[Dict], detected_instances: Optional[List[Instances]] = None,
                  do_postprocess: bool = True, override_boxes: Optional[List[torch.Tensor]] = None):
        return drive(self.inference_g(batched_inputs, detected_instances, do_postprocess, override_boxes))

    def inference_g(self, batched_inputs: List[Dict], detected_instances: Optional[List[Instances]] = None,
                    do_postprocess: bool = True, override_boxes: Optional[List[torch.Tensor]] = None):
        """Generator form of `inference` (utils/pipeline.py): yields a ReadBack wherever the host needs counts;
        `inference` drives it synchronously, `pipeline.run_pipelined` keeps several steps in flight.  Runs under the
        driver's torch.no_grad()."""
        assert not self.training
        if not self._loaded:
            raise RuntimeError("no weights loaded: call load_state_dict()/load_checkpoint() first")
        # (the conv precision travels with the layers' weights - ops.native.Routing - so steps of models of different
        #  precision can interleave segment by segment, or run on different host threads, without any switching here)
        return (yield from self._inference_body_g(batched_inputs, detected_instances, do_postprocess, override_boxes))

    def _trunk(self, nhwc4: torch.Tensor, hw: torch.Tensor) -> Dict[str, torch.Tensor]:
        """padded batch -> FPN features, RPN proposals and the box head's detections: every shape here follows from
        (N, Hp, Wp) alone and nothing reads back to the host (scripts/exp_hipgraph_trunk.py captures exactly this function
        into a hipGraph: on ROCm 7.2 the replay is 3-7 % SLOWER than the ~100 stream launches it replaces - DESIGN.md)."""
        feats = self.backbone.forward_nhwc(nhwc4)
        rpn_in = [feats[f] for f in self.proposal_generator.in_features]
        pboxes, plogits, pcounts = self.proposal_generator.forward_batched(rpn_in, hw)
        ob, os_, oi, orient, oc = self.roi_heads.box_branch_detections(feats, pboxes, pcounts, hw)
        out = {"pboxes": pboxes, "plogits": plogits, "pcounts": pcounts, "ob": ob, "os": os_, "oi": oi, "oc": oc}
        if orient is not None:
            out["orient"] = orient
        out.update({"feat." + k: v for k, v in feats.items()})
        return out

    def _inference_body_g(self, batched_inputs, detected_instances, do_postprocess, override_boxes):
        if detected_instances is None:
            sizes = [(int(x["image"].shape[-2]), int(x["image"].shape[-1])) for x in batched_inputs]
            hw_new = K.upload(sizes, torch.int32, self._device)
            images = self.preprocess_image(batched_inputs)
            t = self._trunk(images.nhwc4, hw_new)
            feats = {k[5:]: v for k, v in t.items() if k.startswith("feat.")}
            det = yield from self.roi_heads.forward_batched_g(
                images.nhwc4, feats, t["pboxes"], t["pcounts"], images.image_sizes, override_boxes=override_boxes,
                box_out=(t["ob"], t["os"], t["oi"], t.get("orient"), t["oc"]))
            det.proposals = (t["pboxes"], t["plogits"], t["pcounts"])      # padded RPN output of this step (parity tests)
            if do_postprocess:
                return (yield from self._postprocess_batched_g(det, batched_inputs, images.image_sizes))
            yield from self._resolve_handoff_g(det)      # (no later read-back on this path: the status gets its own)
            self.last_batch = det                # convenience for the synchronous API only; pipelined callers use .batch
            out = StepOutput(det.to_instances())
            out.batch = det
            return out
        images = self.preprocess_image(batched_inputs)
        feats = self.backbone.forward_nhwc(images.nhwc4)
        detected_instances = [x.to(self._device) for x in detected_instances]
        results = self.roi_heads._recognize_into(images.nhwc4, feats, detected_instances)
        if do_postprocess:
            return self._postprocess(results, batched_inputs, images.image_sizes)
        return results

    # small-box filter of GlassRCNN._postprocess; the stock d2 GeneralizedRCNN has none
    _filter_small = False
    _min_box_dim = 0.0

    @staticmethod
    def _resolve_handoff_g(det):
        """the persistent recurrent kernels' fail-safe for callers WITHOUT a later read-back: read the recognizer call's
        hand-off status now; if a wait gave up, det.text is re-computed on the step kernels (ops.native.HandoffGuard)"""
        guard, det.handoff = getattr(det, "handoff", None), None
        if guard is not None:
            again = guard.resolve((yield ReadBack(guard.status))[0][0])
            if again is not None:
                det.text = again

    def _postprocess_batched(self, det, batched_inputs, image_sizes):
        return drive(self._postprocess_batched_g(det, batched_inputs, image_sizes))

    def _postprocess_batched_g(self, det, batched_inputs, image_sizes):
        """`_postprocess` for the whole step in one kernel (ops.native.detections_finalize) + one read of
        the N surviving counts; per-image Instances are views of the padded outputs."""
        from ..fusion.recognizers_hybrid_head import BatchedDetections
        N = len(image_sizes)
        sxy, ohw, out_sizes = [], [], []
        for inp, image_size in zip(batched_inputs, image_sizes):
            height = inp.get("height", image_size[0])
            width = inp.get("width", image_size[1])
            sxy.append([float(width) / image_size[1], float(height) / image_size[0]])
            ohw.append([int(height), int(width)])
            out_sizes.append((height, width))
        dev = self._device
        scale_xy = K.upload(sxy, torch.float32, dev)
        out_hw = K.upload(ohw, torch.int32, dev)
        roi_start = K.upload(det.roi_start_host, torch.int32, dev) if (det.text is not None or det.masks is not None) else None
        ob, os_, oo, ot, oc = K.detections_finalize(det.boxes, det.scores, det.orient, det.text, det.counts_dev, roi_start,
                                                    scale_xy, out_hw, float(self._min_box_dim), self._filter_small)
        guard, det.handoff = getattr(det, "handoff", None), None
        om = None
        if det.masks is not None:
            # the same ordered compaction for the mask rows (the kernel's "text" slot carries any per-RoI payload)
            _, _, _, om, _ = K.detections_finalize(det.boxes, det.scores, None, det.masks[:, 0].contiguous(), det.counts_dev,
                                                   roi_start, scale_xy, out_hw, float(self._min_box_dim), self._filter_small)
        if guard is None:
            counts = (yield ReadBack(oc))[0].tolist()
        else:
            # the recognizer's hand-off status rides on the read-back of the surviving counts (no extra synchronisation); if a
            # persistent kernel gave up a wait, the text rows are re-computed on the step kernels and finalized again
            host = yield ReadBack(oc, guard.status)
            counts = host[0].tolist()
            again = guard.resolve(host[1][0])
            if again is not None:
                det.text = again
                ob, os_, oo, ot, oc = K.detections_finalize(det.boxes, det.scores, det.orient, det.text, det.counts_dev, roi_start,
                                                            scale_xy, out_hw, float(self._min_box_dim), self._filter_small)
                counts = (yield ReadBack(oc))[0].tolist()
        post = BatchedDetections(ob, os_, oo, oc, counts, out_sizes)
        post.text = ot
        self.last_batch = post
        results = post.to_instances()
        if om is not None:
            # detector_postprocess (post_processor_academic.py:167-176): paste on the scaled boxes; pred_rboxes is
            # scaled once more unless filter_small_boxes' indexing broke its alias with pred_boxes
            for n, r in enumerate(results):
                c = counts[n]
                r.pred_masks = K.paste_rotated_masks(om[n, :c].contiguous(), ob[n, :c].contiguous(), out_sizes[n], 0.5)
                rb = RotatedBoxes(ob[n, :c].clone())
                if not self._filter_small:
                    rb.scale(sxy[n][0], sxy[n][1])
                    rb.clip(out_sizes[n])
                r.pred_rboxes = rb
        out = StepOutput({"instances": r} for r in results)
        out.batch = post
        return out

    def _postprocess(self, instances, batched_inputs, image_sizes):
        out = StepOutput()
        for r, inp, image_size in zip(instances, batched_inputs, image_sizes):
            height = inp.get("height", image_size[0])
            width = inp.get("width", image_size[1])
            out.append({"instances": detector_postprocess(r, height, width)})
        return out


@META_ARCH_REGISTRY.register()
class GlassRCNN(GeneralizedRCNN):
    def __init__(self, cfg):
        super().__init__(cfg)
        pp = cfg.POST_PROCESSING
        self.post_processor = build_post_processor(cfg)
        self.inflate_ratio = pp.INFLATE_RATIO if hasattr(pp, "INFLATE_RATIO") else None
        self.transcript_filtering = pp.TRANSCRIPT_FILTERING if hasattr(pp, "TRANSCRIPT_FILTERING") else None
        self.filter_small_boxes = pp.MIN_BOX_DIMENSION if hasattr(pp, "MIN_BOX_DIMENSION") else None
        self.drop_overlapping_boxes = pp.DROP_OVERLAPPING if hasattr(pp, "DROP_OVERLAPPING") else None
        self.ioa_threshold = pp.IOA_THRESHOLD if hasattr(pp, "IOA_THRESHOLD") else None
        self.valid_score = cfg.INFERENCE_TH_TEST if hasattr(cfg, "INFERENCE_TH_TEST") else 0
        self._filter_small = bool(self.filter_small_boxes)
        self._min_box_dim = float(self.post_processor.min_box_dim)

    def _postprocess(self, instances, batched_inputs, image_sizes):
        out = []
        for r, inp, image_size in zip(instances, batched_inputs, image_sizes):
            height = inp.get("height", image_size[0])
            width = inp.get("width", image_size[1])
            if self.filter_small_boxes:
                r = self.post_processor.filter_small_boxes(r)
            if self.inflate_ratio:
                r = self.post_processor.resize_boxes(r, self.inflate_ratio)
            if self.drop_overlapping_boxes:
                r = self.post_processor.drop_overlapping_boxes(r, self.ioa_threshold, self.valid_score)
            out.append({"instances": detector_postprocess(r, height, width)})
        return out

    def _postprocess_batched_g(self, det, batched_inputs, image_sizes):
        # INFLATE_RATIO / DROP_OVERLAPPING (set by no shipped YAML; the eval CLI pins DROP_OVERLAPPING False) are
        # per-image index logic: take the list-wise path (reference order of operations) instead of the fused kernel
        if self.inflate_ratio or self.drop_overlapping_boxes:
            self.last_batch = None
            yield from self._resolve_handoff_g(det)
            return self._postprocess(det.to_instances(), batched_inputs, image_sizes)
        return (yield from super()._postprocess_batched_g(det, batched_inputs, image_sizes))


def build_model(cfg) -> torch.nn.Module:
    """Work-alike of `detectron2.modeling.build_model`: META_ARCH_REGISTRY lookup by cfg name."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
