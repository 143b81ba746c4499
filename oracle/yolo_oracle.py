"""ORACLE (test infrastructure, never shipped, never imported by wtracker_amd/).

PARITY UNPINNED: the detector arithmetic of the reference lives in the third-party `ultralytics`
package (bare, un-pinned dependency: reference pyproject.toml:24, requirements.yaml:19; package
version 2024.06.15 suggests ultralytics ~8.2.x) which is neither vendored under /root/reference nor
installed here, and the trained weights are missing (.MISSING_LARGE_BLOBS:6-7).  The reference
has no tests or golden vectors for this path.  This file therefore RESTATES the published YOLOv8
algorithm in PyTorch-CPU fp32 and anchors it on the reference's own call site:

  YoloController.predict     wtracker/sim/sim_controllers/yolo_controller.py:64-90
     gray -> BGR replicate (:68-69); YOLO.predict(source, max_det=1, imgsz, conf) (:72-78);
     no box -> NaN x 4 (:84-85); xyxy -> xywh (:87); np.stack (:90)
  hyper-parameters           yolo/yolo_train_config.yaml:13 (imgsz 384), :27 (single_cls),
                             :49 (iou 0.7), :51 (half False => fp32), :61 (class-aware NMS)

Restated third-party steps (public YOLOv8 / ultralytics behaviour):
  LetterBox(auto=True) + BGR->RGB + HWC->CHW + /255;  DetectionModel forward (Conv = conv+BN+SiLU
  fused, C2f, SPPF, nearest Upsample, Concat, Detect with DFL);  dist2bbox(xywh) * stride;
  sigmoid;  non_max_suppression(conf, iou, max_det);  scale_boxes + clip.

Self-checks that pin the restatement without an external oracle (tests/test_oracle_yolo.py):
conv count 63, 14.216 GMAC @640^2 (nc=1), 28.60 GFLOP (nc=80), 11,166,560 parameters (nc=80, the
public model card), zero box head -> every box is 15*stride wide centred on its anchor, identity
letterbox when H = W = imgsz.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

STRIDES = (8, 16, 32)
REG_MAX = 16


class YoloOracle:
    def __init__(self, weights: dict, dims: dict):
        """weights: name -> (W [cout,kh,kw,cin] fp32, b [cout]); dims: yolo_spec.model_dims(...)."""
        self.w = {k: (torch.from_numpy(np.ascontiguousarray(w)).permute(0, 3, 1, 2).contiguous(), torch.from_numpy(np.ascontiguousarray(b)))
                  for k, (w, b) in weights.items()}
        self.dims = dims

    # ---- modules -------------------------------------------------------------------------------
    def conv(self, name, x, stride=1, act=True):
        w, b = self.w[name]
        y = F.conv2d(x, w, b, stride=stride, padding=w.shape[2] // 2)
        return F.silu(y) if act else y

    def c2f(self, p, x, n, shortcut):
        y = list(self.conv(p + ".cv1", x).chunk(2, 1))
        for i in range(n):
            t = self.conv(f"{p}.m.{i}.cv2", self.conv(f"{p}.m.{i}.cv1", y[-1]))
            y.append(y[-1] + t if shortcut else t)
        return self.conv(p + ".cv2", torch.cat(y, 1))

    def sppf(self, x):
        y = [self.conv("model.9.cv1", x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv("model.9.cv2", torch.cat(y, 1))

    def forward(self, x: torch.Tensor):
        """x [B,3,H,W] RGB float in [0,1] -> (box logits [B,A,64], cls logits [B,A,nc]), anchors level-major, x fastest."""
        n = self.dims["n"]
        x0 = self.conv("model.0", x, 2)
        x1 = self.conv("model.1", x0, 2)
        x2 = self.c2f("model.2", x1, n[0], True)
        x3 = self.conv("model.3", x2, 2)
        x4 = self.c2f("model.4", x3, n[1], True)
        x5 = self.conv("model.5", x4, 2)
        x6 = self.c2f("model.6", x5, n[2], True)
        x7 = self.conv("model.7", x6, 2)
        x8 = self.c2f("model.8", x7, n[3], True)
        x9 = self.sppf(x8)
        x11 = torch.cat([F.interpolate(x9, scale_factor=2, mode="nearest"), x6], 1)
        x12 = self.c2f("model.12", x11, n[3], False)
        x14 = torch.cat([F.interpolate(x12, scale_factor=2, mode="nearest"), x4], 1)
        x15 = self.c2f("model.15", x14, n[3], False)
        x17 = torch.cat([self.conv("model.16", x15, 2), x12], 1)
        x18 = self.c2f("model.18", x17, n[3], False)
        x20 = torch.cat([self.conv("model.19", x18, 2), x9], 1)
        x21 = self.c2f("model.21", x20, n[3], False)
        boxes, clss = [], []
        for i, f in enumerate((x15, x18, x21)):
            b = self.conv(f"model.22.cv2.{i}.2", self.conv(f"model.22.cv2.{i}.1", self.conv(f"model.22.cv2.{i}.0", f)), act=False)
            c = self.conv(f"model.22.cv3.{i}.2", self.conv(f"model.22.cv3.{i}.1", self.conv(f"model.22.cv3.{i}.0", f)), act=False)
            B = b.shape[0]
            boxes.append(b.reshape(B, 64, -1).permute(0, 2, 1))
            clss.append(c.reshape(B, c.shape[1], -1).permute(0, 2, 1))
        return torch.cat(boxes, 1).contiguous(), torch.cat(clss, 1).contiguous()


class UnfusedYoloOracle(YoloOracle):
    """The same graph on an UN-FUSED ultralytics-style state dict (what a real `yolov8s_trained.pt` holds): every `Conv` is
    Conv2d(no bias) -> BatchNorm2d(eps 1e-3, running statistics) -> SiLU, the six Detect output convs are plain Conv2d with
    bias.  Checks tools/convert_ultralytics.py end to end: nothing here folds anything."""

    BN_EPS = 1e-3  # ultralytics Conv: nn.BatchNorm2d(c2, eps=0.001, momentum=0.03)

    def __init__(self, state_dict: dict, dims: dict):
        self.sd = {k: torch.as_tensor(np.asarray(v)).float() for k, v in state_dict.items()}
        self.dims = dims

    def conv(self, name, x, stride=1, act=True):
        sd = self.sd
        if name + ".conv.weight" in sd:
            w = sd[name + ".conv.weight"]
            y = F.conv2d(x, w, None, stride=stride, padding=w.shape[2] // 2)
            y = F.batch_norm(y, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                             training=False, eps=self.BN_EPS)
            assert act
            return F.silu(y)
        w = sd[name + ".weight"]
        assert not act
        return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=w.shape[2] // 2)


# ---- pre / post processing ----------------------------------------------------------------------
def letterbox_geometry(h: int, w: int, imgsz: int, stride: int = 32):
    """ultralytics LetterBox(auto=True, center=True): (net_h, net_w, new_h, new_w, top, left)."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (imgsz - new_w) % stride, (imgsz - new_h) % stride
    dw, dh = dw / 2, dh / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    bottom, right = int(round(dh + 0.1)), int(round(dw + 0.1))
    return new_h + top + bottom, new_w + left + right, new_h, new_w, top, left


def resize_bilinear_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(INTER_LINEAR) for uint8: half-pixel centres, edge clamp, 11-bit fixed-point weights."""
    h, w = img.shape[:2]
    im = img.reshape(h, w, -1).astype(np.int64)

    def coeffs(n_dst, n_src):
        s = np.float32(n_src) / np.float32(n_dst)
        f = ((np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * s - np.float32(0.5)).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0[lo], f[lo] = 0, 0
        hi = i0 >= n_src - 1
        i0[hi], f[hi] = n_src - 1, 0
        i1 = np.minimum(i0 + 1, n_src - 1)
        a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return i0, i1, 2048 - a1, a1

    x0, x1, ax0, ax1 = coeffs(new_w, w)
    y0, y1, ay0, ay1 = coeffs(new_h, h)
    rows = im[:, x0, :] * ax0[None, :, None] + im[:, x1, :] * ax1[None, :, None]  # horizontal pass, int
    r0, r1 = rows[y0], rows[y1]
    v = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out.reshape(new_h, new_w, *img.shape[2:])


def preprocess(frames, imgsz: int) -> tuple:
    """list of HxW / HxWx3(BGR) uint8 frames of one shape -> (tensor [B,3,net_h,net_w] RGB /255, (h, w))."""
    out = []
    h, w = frames[0].shape[:2]
    net_h, net_w, new_h, new_w, top, left = letterbox_geometry(h, w, imgsz)
    for f in frames:
        if f.ndim == 2:
            f = np.repeat(f[..., None], 3, axis=2)  # cv.COLOR_GRAY2BGR
        if (new_h, new_w) != (h, w):
            f = resize_bilinear_u8(f, new_h, new_w)
        canvas = np.full((net_h, net_w, 3), 114, dtype=np.uint8)
        canvas[top : top + new_h, left : left + new_w] = f
        out.append(canvas[..., ::-1].transpose(2, 0, 1))  # BGR -> RGB, HWC -> CHW
    x = torch.from_numpy(np.ascontiguousarray(np.stack(out))).float() / 255
    return x, (h, w)


def decode(box_logits: torch.Tensor, cls_logits: torch.Tensor, net_hw: tuple):
    """Detect inference path: DFL + dist2bbox(xywh=True) * stride, sigmoid.  -> (xywh [B,A,4], scores [B,A,nc])."""
    B, A, _ = box_logits.shape
    aps, strs = [], []
    for s in STRIDES:
        hh, ww = net_hw[0] // s, net_hw[1] // s
        sx = torch.arange(ww, dtype=torch.float32) + 0.5
        sy = torch.arange(hh, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        aps.append(torch.stack((xx, yy), -1).view(-1, 2))
        strs.append(torch.full((hh * ww, 1), float(s)))
    anchors, strides = torch.cat(aps), torch.cat(strs)
    dist = box_logits.view(B, A, 4, REG_MAX).softmax(3) @ torch.arange(REG_MAX, dtype=torch.float32)
    lt, rb = dist[..., :2], dist[..., 2:]
    x1y1, x2y2 = anchors - lt, anchors + rb
    xywh = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), -1) * strides
    return xywh, cls_logits.sigmoid()


def _iou_one_to_many(b, bs):
    x1, y1 = torch.maximum(b[0], bs[:, 0]), torch.maximum(b[1], bs[:, 1])
    x2, y2 = torch.minimum(b[2], bs[:, 2]), torch.minimum(b[3], bs[:, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    a = (b[2] - b[0]) * (b[3] - b[1])
    as_ = (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])
    return inter / (a + as_ - inter)


def nms(xywh: torch.Tensor, scores: torch.Tensor, conf: float, iou: float, max_det: int, max_wh: float = 7680.0):
    """non_max_suppression for ONE image (class-aware, best class per anchor).  Returns
    (boxes xyxy [k,4], conf [k], cls [k], anchor index [k]).  Ties resolve to the lower anchor index
    (stable descending sort), as torchvision's CPU nms does."""
    s, j = scores.max(1)
    cand = torch.nonzero(s > conf).flatten()
    if cand.numel() == 0:
        z = torch.zeros(0)
        return torch.zeros(0, 4), z, z, torch.zeros(0, dtype=torch.long)
    b = xywh[cand]
    xy, wh = b[:, :2], b[:, 2:]
    xyxy = torch.cat((xy - wh / 2, xy + wh / 2), 1)
    sc, cl = s[cand], j[cand]
    order = torch.sort(sc, descending=True, stable=True).indices[:30000]
    xyxy, sc, cl, cand = xyxy[order], sc[order], cl[order], cand[order]
    off = xyxy + cl[:, None].float() * max_wh
    keep = []
    alive = torch.ones(len(sc), dtype=torch.bool)
    for i in range(len(sc)):
        if not alive[i]:
            continue
        keep.append(i)
        if len(keep) >= max_det:
            break
        rest = torch.nonzero(alive).flatten()
        rest = rest[rest > i]
        if rest.numel():
            alive[rest[_iou_one_to_many(off[i], off[rest]) > iou]] = False
    k = torch.tensor(keep, dtype=torch.long)
    return xyxy[k], sc[k], cl[k], cand[k]


def scale_boxes(net_hw, xyxy: torch.Tensor, img_hw):
    gain = min(net_hw[0] / img_hw[0], net_hw[1] / img_hw[1])
    pad_x = round((net_hw[1] - img_hw[1] * gain) / 2 - 0.1)
    pad_y = round((net_hw[0] - img_hw[0] * gain) / 2 - 0.1)
    out = xyxy.clone()
    out[:, [0, 2]] -= pad_x
    out[:, [1, 3]] -= pad_y
    out /= gain
    out[:, [0, 2]] = out[:, [0, 2]].clamp(0, img_hw[1])
    out[:, [1, 3]] = out[:, [1, 3]].clamp(0, img_hw[0])
    return out


def postprocess(box_logits, cls_logits, net_hw, img_hw, conf=0.1, iou=0.7, max_det=1):
    """-> (xywh [B,4] with NaN rows as np.float32/64 like the reference, conf [B], anchor [B])."""
    xywh, scores = decode(box_logits, cls_logits, net_hw)
    rows, confs, anchors = [], [], []
    for n in range(xywh.shape[0]):
        xyxy, sc, _, idx = nms(xywh[n], scores[n], conf, iou, max_det)
        if len(sc) == 0:
            rows.append(np.full([4], np.nan))
            confs.append(0.0)
            anchors.append(-1)
            continue
        b = scale_boxes(net_hw, xyxy[:1], img_hw)[0].numpy()
        rows.append(np.array([b[0], b[1], b[2] - b[0], b[3] - b[1]], dtype=np.float32))  # to_xywh
        confs.append(float(sc[0]))
        anchors.append(int(idx[0]))
    return np.stack(rows, 0), np.asarray(confs, dtype=np.float32), np.asarray(anchors, dtype=np.int32)


def predict(model: YoloOracle, frames, imgsz: int = 640, conf: float = 0.1, iou: float = 0.7, max_det: int = 1):
    """YoloController.predict on the CPU restatement."""
    assert len(frames) > 0
    with torch.no_grad():
        x, img_hw = preprocess(list(frames), imgsz)
        box, cls = model.forward(x)
        return postprocess(box, cls, tuple(x.shape[2:]), img_hw, conf, iou, max_det)
