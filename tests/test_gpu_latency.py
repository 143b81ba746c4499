"""GPU parity of the LATENCY plan (conv_sk.hip: split-K implicit GEMM + slab-combining pass; wtk_yolo_create_planned) — the plan a handle of
max_batch <= 16 gets, i.e. the reference's own operating point: one call of cycle_frame_num frames and one single-frame call per cycle at
imgsz 384 (yolo_controller.py:96-98,108-109).  Checker: the fp32 CPU restatement (oracle/yolo_oracle.py; parity unpinned: no ultralytics here).

Bars (written here): head logits within 2e-3, boxes within 2e-2 px, survivor index equal — on 256 frames, at B = 1 and B = 15;
and a frame's logits must not depend on the batch it arrives in (bit for bit, within the plan)."""
import os

import numpy as np
import pytest
import torch

from oracle import yolo_oracle as yo
from wtracker_amd import frames as fr
from wtracker_amd import hip
from wtracker_amd import yolo_spec as ys

pytestmark = pytest.mark.gpu

LOGIT_ATOL = 2e-3
BOX_ATOL = 2e-2


def _handle(size, dtype, max_batch=16, plan="latency", nc=1, seed=0, scale="s"):
    w = ys.synthetic_weights(scale, nc, seed=seed)
    depth, width, maxch = ys.SCALES[scale]
    det = hip.HipYolo(w, (size, size), max_batch, dtype=dtype, nc=nc, width=width, depth=depth, max_channels=maxch, plan=plan)
    return w, det, yo.YoloOracle(w, ys.model_dims(width, depth, maxch, nc))


def _oracle(oracle, frames, size, conf=0.1):
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box, cls = oracle.forward(x)
    return box.numpy(), cls.numpy(), yo.postprocess(box, cls, (size, size), hw, conf=conf)


def test_auto_rule_and_explicit_plans(hip_lib, monkeypatch):
    w = ys.synthetic_weights("s", 1, seed=0)
    mk = lambda **kw: hip.HipYolo(w, (128, 128), **kw)
    monkeypatch.delenv("WTK_LATENCY_PLAN", raising=False)
    for dtype, mb, want in (("fp32", 4, "latency"), ("f16x3", 1, "latency"), ("f16x3", 5, "throughput"), ("fp16", 2, "throughput")):
        d = mk(max_batch=mb, dtype=dtype)
        assert d.plan == want, (dtype, mb, d.plan)
        d.close()
    monkeypatch.setenv("WTK_LATENCY_PLAN", "0")  # the variable overrides AUTO ...
    d = mk(max_batch=2, dtype="fp32")
    assert d.plan == "throughput"
    d.close()
    d = mk(max_batch=8, dtype="fp32", plan="latency")  # ... and an explicit plan beats the variable
    assert d.plan == "latency"
    d.close()
    d = mk(max_batch=64, dtype="f16x3", plan="latency")
    assert d.plan == "latency"
    d.close()
    with pytest.raises(hip.WtkError, match="latency plan"):
        mk(max_batch=8, dtype="fp16", plan="latency")


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
@pytest.mark.parametrize("size,B", [(384, 1), (384, 5), (640, 1), (128, 3)])
def test_latency_plan_logits_boxes_and_survivors_match_the_restatement(hip_lib, dtype, size, B):
    _, det, oracle = _handle(size, dtype)
    assert det.plan == "latency"
    frames = fr.diverse_frames(max(B, 4), size, seed=500 + size)[:B]
    box_o, cls_o, (xywh_o, conf_o, anchor_o) = _oracle(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    np.testing.assert_allclose(cls_g, cls_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=BOX_ATOL)
    np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-4)
    det.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_a_frames_logits_do_not_depend_on_its_batch(hip_lib, dtype):
    """The K slicing is a function of the layer alone and the tile shape never enters the arithmetic: the same frame alone, in a batch of 4 and
    in a batch of 15 gives the same head logits bit for bit (what lets provide_movement_vector's single-frame call and _cycle_predict_all's
    cycle batch agree on a frame they both see, yolo_controller.py:96-109)."""
    size = 384
    _, det, _ = _handle(size, dtype)
    frames = fr.diverse_frames(16, size, seed=77)[:15]
    x15, c15, a15 = det.predict_host(frames, conf=0.1)
    b15, k15 = det.debug_head(15)
    for i0, n in ((0, 1), (7, 1), (14, 1), (4, 4), (11, 4)):
        x, c, a = det.predict_host(frames[i0 : i0 + n], conf=0.1)
        b, k = det.debug_head(n)
        np.testing.assert_array_equal(k, k15[i0 : i0 + n])
        np.testing.assert_array_equal(b, b15[i0 : i0 + n])
        np.testing.assert_array_equal(a, a15[i0 : i0 + n])
        np.testing.assert_array_equal(x, x15[i0 : i0 + n])
    det.close()


def test_survivor_index_on_256_frames_at_b15_and_b1(hip_lib):
    """256 frames of 64 seeded tracks at 384^2 through the f16x3 latency handle in calls of 15 (the cycle batch; the last call has one frame) and the
    first 32 again one at a time: every survivor index equals the fp32 restatement's, boxes within 2e-2 px."""
    size, n = 384, 256
    _, det, oracle = _handle(size, "f16x3")
    frames = fr.diverse_frames(n, size, seed=9000)
    res = [det.predict_host(frames[i : i + 15], conf=0.1) for i in range(0, n, 15)]
    xywh, conf, anchor = (np.concatenate([r[k] for r in res]) for k in range(3))
    xo, ao = [], []
    for i in range(0, n, 32):
        _, _, (x, _, a) = _oracle(oracle, frames[i : i + 32], size)
        xo.append(x), ao.append(a)
    xywh_o, anchor_o = np.concatenate(xo), np.concatenate(ao)
    assert (anchor_o >= 0).sum() >= n // 2  # the threshold is exercised on both sides
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=BOX_ATOL)
    for i in range(32):
        x1, _, a1 = det.predict_host(frames[i : i + 1], conf=0.1)
        assert a1[0] == anchor[i]
        np.testing.assert_array_equal(x1[0], xywh[i])
    det.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_latency_and_throughput_plans_agree_within_the_stated_tolerance(hip_lib, dtype):
    """Two plans of the same model: K is summed in a different order, so not bit-identical — logits within the restatement tolerance of each
    other, survivors equal on 30 frames."""
    size, n = 384, 30
    w, lat, _ = _handle(size, dtype)
    depth, width, maxch = ys.SCALES["s"]
    thr = hip.HipYolo(w, (size, size), 16, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan="throughput")
    assert thr.plan == "throughput"
    frames = fr.diverse_frames(32, size, seed=31)[:n]
    for i in range(0, n, 15):
        xl, cl, al = lat.predict_host(frames[i : i + 15], conf=0.1)
        bl, kl = lat.debug_head(15)
        xt, ct, at = thr.predict_host(frames[i : i + 15], conf=0.1)
        bt, kt = thr.debug_head(15)
        np.testing.assert_allclose(kl, kt, rtol=1e-3, atol=LOGIT_ATOL)
        np.testing.assert_allclose(bl, bt, rtol=1e-3, atol=LOGIT_ATOL)
        np.testing.assert_array_equal(al, at)
        np.testing.assert_allclose(xl, xt, rtol=0, atol=BOX_ATOL)
    lat.close(), thr.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_in_kernel_slab_combination_equals_the_two_launch_form(hip_lib, dtype, monkeypatch):
    """The slabs of a split layer are combined by the block of the tile that arrives last (write-through stores, ticket, sc1 loads: conv_sk.hip) — a
    cross-CU hand-off inside one launch.  WTK_SK_FINISH=1 builds the same handle with the combination as a second launch (ordinary kernel-boundary
    visibility).  Both add the slabs in slice order, so every logit must be equal bit for bit; a stale or torn slab read would show here.  Repeated
    with changing frames and batch sizes so that slab addresses are re-used while other work is in flight."""
    size = 384
    w, a, _ = _handle(size, dtype)
    monkeypatch.setenv("WTK_SK_FINISH", "1")
    depth, width, maxch = ys.SCALES["s"]
    b = hip.HipYolo(w, (size, size), 16, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan="latency")
    monkeypatch.delenv("WTK_SK_FINISH")
    pool = fr.diverse_frames(64, size, seed=123)
    rng = np.random.default_rng(0)
    for it in range(24):
        n = (15, 1, 4, 9)[it % 4]
        sel = pool[rng.integers(0, 64, size=n)]
        xa, ca, aa = a.predict_host(sel, conf=0.1)
        ba, ka = a.debug_head(n)
        xb, cb, ab = b.predict_host(sel, conf=0.1)
        bb, kb = b.debug_head(n)
        np.testing.assert_array_equal(ka, kb, err_msg=f"iteration {it}")
        np.testing.assert_array_equal(ba, bb, err_msg=f"iteration {it}")
        np.testing.assert_array_equal(aa, ab)
        np.testing.assert_array_equal(xa, xb)
    a.close(), b.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_walking_all_atoms_in_one_block_equals_one_block_per_atom(hip_lib, dtype, monkeypatch):
    """An output value is defined as ((A_0 + A_1) + ...) over the layer's K atoms.  The launcher picks per call between one block per atom (slabs +
    combination) and one block that walks all atoms and folds them at the atom boundaries (many pixels: no slabs), and between four tile shapes.
    WTK_SK_FORM / WTK_SK_TILE force each choice for every layer: all of them must give the same logits bit for bit — that is what makes the choice
    free to depend on the batch."""
    size, B = 384, 5
    frames = fr.diverse_frames(8, size, seed=321)[:B]
    ref = None
    for form, tile in (("-1", "-1"), ("0", "-1"), ("1", "-1"), ("0", "2"), ("1", "3"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("WTK_SK_FORM", form)  # read when the handle is created
        monkeypatch.setenv("WTK_SK_TILE", tile)
        _, det, _ = _handle(size, dtype)
        x, c, a = det.predict_host(frames, conf=0.1)
        b, k = det.debug_head(B)
        det.close()
        if ref is None:
            ref = (x, a, b, k)
            continue
        np.testing.assert_array_equal(k, ref[3], err_msg=f"form {form} tile {tile}")
        np.testing.assert_array_equal(b, ref[2], err_msg=f"form {form} tile {tile}")
        np.testing.assert_array_equal(a, ref[1])
        np.testing.assert_array_equal(x, ref[0])


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
@pytest.mark.parametrize("size,B", [(384, 1), (384, 15), (640, 2), (128, 3)])
def test_grouped_level_launches_equal_one_launch_per_conv(hip_lib, dtype, size, B, monkeypatch):
    """Round 6: a latency-plan handle launches the split-K convs of one dependency level (a Detect tower's box and class convs, a PAN layer next to the
    tower of the feature map before it) as ONE grid on the caller's stream (csrc/wtk_plan.hip: sk_schedule, conv_sk.hip: launch_conv_sk_group).  Grouping
    picks one tile per launch and a form per member — neither enters the arithmetic — so EVERY conv tensor and every row must equal the
    one-launch-per-conv handle (WTK_SK_GROUP=0) bit for bit."""
    frames = fr.diverse_frames(max(B, 4), size, seed=77 + size)[:B]
    _, grouped, _ = _handle(size, dtype)
    monkeypatch.setenv("WTK_SK_GROUP", "0")
    _, single, _ = _handle(size, dtype)
    xg, cg, ag = grouped.predict_host(frames, conf=0.1)
    xs, cs, as_ = single.predict_host(frames, conf=0.1)
    np.testing.assert_array_equal(ag, as_)
    np.testing.assert_array_equal(xg, xs)
    np.testing.assert_array_equal(cg, cs)
    depth, width, maxch = ys.SCALES["s"]
    n_convs = len(hip.yolo_conv_table(width, depth, maxch, 1))
    seen = 0
    for ci in range(n_convs):
        try:
            tg = grouped.debug_tensor(ci, B)
        except hip.WtkError:
            continue  # (a class tower's first conv rides in the box tower's op)
        np.testing.assert_array_equal(tg, single.debug_tensor(ci, B), err_msg=f"conv {ci}")
        seen += 1
    assert seen >= 57
    grouped.close(), single.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_slab_hand_off_holds_beside_another_handles_kernels(hip_lib, dtype, monkeypatch):
    """the in-kernel slab hand-off (sc1 stores, ticket, sc1 loads: the form MI355X_MICROARCH.md measured for one workgroup per CU) must also
    hold when blocks of OTHER kernels share the CUs — the deferred track log runs a cycle batch on lane 1 beside the single-frame call.  A second handle
    streams large batches on another stream (uneven load: its launches of 140-270 blocks come and go) while the handle under test runs single frames;
    every conv tensor of every round is compared, word for word, with the two-launch form (WTK_SK_FINISH=1: plain stores, slabs combined by a second
    launch — no in-kernel hand-off at all)."""
    size = 384
    frames = fr.diverse_frames(8, size, seed=11)
    dev = torch.device("cuda", 0)
    _, det, _ = _handle(size, dtype, max_batch=4)
    monkeypatch.setenv("WTK_SK_FINISH", "1")
    _, ref, _ = _handle(size, dtype, max_batch=4)
    monkeypatch.delenv("WTK_SK_FINISH")
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    noise = hip.HipYolo(w, (size, size), 16, dtype=dtype, nc=1, width=width, depth=depth, max_channels=maxch, plan="throughput")
    nf = torch.from_numpy(fr.diverse_frames(16, size, seed=12)).to(dev)
    no, na = torch.empty((16, 4), dtype=torch.float32, device=dev), torch.empty((16,), dtype=torch.int32, device=dev)
    ns = torch.cuda.Stream(device=dev)
    n_convs = len(hip.yolo_conv_table(width, depth, maxch, 1))
    expect = {}
    for r in range(4):  # the reference rounds run alone
        ref.predict_host(frames[r : r + 1], conf=0.1)
        expect[r] = [ref.debug_tensor(ci, 1) if _has(ref, ci) else None for ci in range(n_convs)]
    for it in range(24):
        r = it % 4
        for k in range(3 + it % 5):  # uneven: a different amount of foreign work in flight every round
            noise.predict(nf, 5 + (it + k) % 11, size, size, 1, no, None, na, conf=0.1, stream=ns.cuda_stream)
        det.predict_host(frames[r : r + 1], conf=0.1)  # host stream of the handle: runs beside the noise stream
        for ci in range(n_convs):
            if expect[r][ci] is not None:
                np.testing.assert_array_equal(det.debug_tensor(ci, 1), expect[r][ci], err_msg=f"round {it} conv {ci}")
    ns.synchronize()
    det.close(), ref.close(), noise.close()


def _has(det, ci):
    try:
        det.debug_tensor(ci, 1)
        return True
    except hip.WtkError:
        return False


def test_replayed_capture_for_caller_buffers_equals_eager(hip_lib, monkeypatch):
    """With WTK_GRAPH_VIEWS=1 a latency handle replays a captured hipGraph for caller buffers from the third call with the same argument set on (seen
    once -> captured -> replayed); the rows must be the eager rows, and new frame CONTENT in the same buffer must be honoured by the replay."""
    size, B = 384, 15
    monkeypatch.setenv("WTK_GRAPH_VIEWS", "1")
    _, det, _ = _handle(size, "f16x3")
    dev = torch.device("cuda", 0)
    fa = torch.from_numpy(fr.diverse_frames(16, size, seed=1)[:B]).to(dev)
    fb = torch.from_numpy(fr.diverse_frames(16, size, seed=2)[:B]).to(dev)
    buf = torch.empty_like(fa)
    out = torch.empty((B, 4), dtype=torch.float32, device=dev)
    an = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    rows = []
    with torch.cuda.stream(st):
        for src in (fa, fa, fa, fb, fa):
            buf.copy_(src)
            det.predict(buf, B, size, size, 1, out, None, an, conf=0.1, stream=st.cuda_stream)
            st.synchronize()
            rows.append((out.cpu().numpy().copy(), an.cpu().numpy().copy()))
    for k in (1, 2, 4):
        np.testing.assert_array_equal(rows[k][0], rows[0][0])
        np.testing.assert_array_equal(rows[k][1], rows[0][1])
    xb, _, ab = det.predict_host(fb.cpu().numpy(), conf=0.1)
    np.testing.assert_array_equal(rows[3][1], ab)
    np.testing.assert_array_equal(rows[3][0], xb)
    assert not np.array_equal(rows[3][1], rows[0][1])
    det.close()


def test_opt_in_forked_capture_on_a_large_handle_equals_eager(hip_lib, monkeypatch):
    """Replayed captures are opt-in since round 6 (WTK_GRAPH=1).  A handle of more than 16 frames keeps its P3 / P4 towers on the process-wide side streams,
    so its capture FORKS (the form round 5 replayed by default): captured the second time an argument set is met, replayed afterwards — rows and head
    logits must equal the eager handle's, and the handle must come apart cleanly (execs before events before streams: csrc/wtk_plan.hip)."""
    size, B = 256, 8
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    mk = lambda: hip.HipYolo(w, (size, size), 32, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch, plan="throughput")
    eager = mk()
    monkeypatch.setenv("WTK_GRAPH", "1")
    graph = mk()
    dev = torch.device("cuda", 0)
    f = torch.from_numpy(fr.diverse_frames(8, size, seed=9)[:B]).to(dev)
    out = [torch.empty((B, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    an = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(2)]
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(4):  # eager, captured, replayed, replayed
            graph.predict(f, B, size, size, 1, out[0], None, an[0], conf=0.1, stream=st.cuda_stream)
        st.synchronize()
        bg, kg = graph.debug_head(B)
        eager.predict(f, B, size, size, 1, out[1], None, an[1], conf=0.1, stream=st.cuda_stream)
        st.synchronize()
        be, ke = eager.debug_head(B)
    np.testing.assert_array_equal(kg, ke)
    np.testing.assert_array_equal(bg, be)
    np.testing.assert_array_equal(an[0].cpu().numpy(), an[1].cpu().numpy())
    np.testing.assert_array_equal(out[0].cpu().numpy(), out[1].cpu().numpy())
    xg = graph.predict_host(f.cpu().numpy(), conf=0.1)  # host entry point: own staging buffers, captured at the first call
    xg2 = graph.predict_host(f.cpu().numpy(), conf=0.1)
    np.testing.assert_array_equal(xg[0], xg2[0])
    np.testing.assert_array_equal(xg[0], out[1].cpu().numpy())
    graph.close(), eager.close()


def test_two_host_threads_a_handle_each(hip_lib):
    """README, "Threads": handles driven from different host threads (ctypes releases the GIL) are safe.  Two threads, a handle each — one latency-plan
    handle (one stream, grouped launches), one handle of 32 frames (side streams, shared process-wide, under the library's lock) — run interleaved
    calls; every call's rows and head logits equal the ones the same handle produced alone."""
    import threading

    size = 256
    frames = fr.diverse_frames(8, size, seed=31)
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    dets = [hip.HipYolo(w, (size, size), 4, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch, plan="latency"),
            hip.HipYolo(w, (size, size), 32, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch, plan="throughput")]
    batches = [[frames[i % 8 : i % 8 + 1] for i in range(12)], [frames[: 1 + i % 8] for i in range(12)]]
    alone = [[d.predict_host(b, conf=0.1) for b in bs] for d, bs in zip(dets, batches)]
    got, errs = [[], []], []

    def work(k):
        try:
            for b in batches[k]:
                got[k].append(dets[k].predict_host(b, conf=0.1))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for k in range(2):
        assert len(got[k]) == len(alone[k])
        for (xa, ca, aa), (xg, cg, ag) in zip(alone[k], got[k]):
            np.testing.assert_array_equal(ag, aa)
            np.testing.assert_array_equal(xg, xa)
            np.testing.assert_array_equal(cg, ca)
    for d in dets:
        d.close()


def test_dynamic_batch_on_a_latency_handle(hip_lib):
    size, B = 128, 8
    _, det, _ = _handle(size, "f16x3")
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(fr.diverse_frames(8, size, seed=5)).to(dev)
    out = torch.full((B, 4), -1.0, dtype=torch.float32, device=dev)
    an = torch.full((B,), -7, dtype=torch.int32, device=dev)
    det.predict(frames, B, size, size, 1, out, None, an, conf=0.1)
    torch.cuda.synchronize()
    full = an.cpu().numpy().copy()
    n_dev = torch.tensor([3], dtype=torch.int32, device=dev)
    det.set_dynamic_batch(n_dev)
    det.predict(frames, B, size, size, 1, out, None, an, conf=0.1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(an.cpu().numpy()[:3], full[:3])
    det.set_dynamic_batch(None)
    det.close()


# ---- range guard of the fp16-storage modes (include/wtk_hip.h: wtk_yolo_status)
def _scaled_weights(name, factor, seed=0):
    w = {k: (a.copy(), b.copy()) for k, (a, b) in ys.synthetic_weights("s", 1, seed=seed).items()}
    w[name] = (w[name][0] * np.float32(factor), w[name][1])
    return w


@pytest.mark.parametrize("plan", ["latency", "throughput"])
def test_activation_overflow_raises_the_sticky_flag_in_fp16_modes_and_not_in_fp32(hip_lib, plan):
    """One conv's weights x 3e4: its outputs (O(1 .. 10) before) leave the fp16 range.  fp16 / f16x3 handles must raise WTK_STATUS_NONFINITE on the first
    call and keep it until cleared; the fp32 handle computes the (large but finite) values and stays clean."""
    size, B = 128, 2
    w = _scaled_weights("model.6.cv2", 3e4)
    depth, width, maxch = ys.SCALES["s"]
    frames = fr.diverse_frames(4, size, seed=3)[:B]
    for dtype in ("f16x3", "fp16", "fp32"):
        if dtype == "fp16" and plan == "latency":
            continue
        det = hip.HipYolo(w, (size, size), 4, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan=plan)
        assert det.status() == 0
        det.predict_host(frames, conf=0.1)
        if dtype == "fp32":
            assert det.status() == 0, dtype
        else:
            assert det.status() & hip.STATUS_NONFINITE, dtype
            assert det.status(clear=True) & hip.STATUS_NONFINITE  # sticky until cleared
            assert det.status() == 0
        det.close()
    clean = hip.HipYolo(ys.synthetic_weights("s", 1, seed=0), (size, size), 4, dtype="f16x3", width=width, depth=depth, max_channels=maxch, plan=plan)
    clean.predict_host(frames, conf=0.1)
    assert clean.status() == 0
    clean.close()


def test_weights_outside_the_fp16_range_are_refused_at_create(hip_lib):
    depth, width, maxch = ys.SCALES["s"]
    w = _scaled_weights("model.22.cv2.0.2", 1e6)  # a linear Detect output conv: |w| ~ 1e5 after the fold
    for dtype in ("f16x3", "fp16"):
        with pytest.raises(hip.WtkError, match="outside the fp16 range"):
            hip.HipYolo(w, (128, 128), 2, dtype=dtype, width=width, depth=depth, max_channels=maxch)
    det = hip.HipYolo(w, (128, 128), 2, dtype="fp32", width=width, depth=depth, max_channels=maxch)  # the reference's precision takes them
    det.close()


def test_controller_raises_on_overflow(hip_lib, tmp_path):
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig

    path = str(tmp_path / "big.wtk")
    ys.save_weights(path, _scaled_weights("model.6.cv2", 3e4), "s", 1)
    ec = ExperimentConfig("x", 20, 60, (256, 256), 90, (128, 128))
    tc = TimingConfig(ec, 100, 40, 50, (1.4, 1.4), (0.32, 0.32))
    frames = list(fr.diverse_frames(4, 128, seed=3)[:2])
    ctrl = HipYoloController(tc, YoloConfig(model_path=path, pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="f16x3", max_batch=4))
    with pytest.raises(hip.WtkError, match="fp16 range"):
        ctrl.predict(frames)
    ok = HipYoloController(tc, YoloConfig(model_path=path, pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="fp32", max_batch=4))
    assert ok.predict(frames).shape == (2, 4)


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_small_throughput_handle_runs_its_smallest_maps_on_the_split_k_kernel(hip_lib, dtype, monkeypatch):
    """A throughput-plan handle of max_batch <= 16 (the controller's cycle batch: 9 / 15 frames at imgsz 384) runs the layers whose whole batch is at most
    4 096 output pixels — the 12 x 12 maps — on conv_sk_kernel (csrc/wtk_plan.hip: sk_mixed).  Same bars as every reference-precision path: logits 2e-3,
    boxes 2e-2 px, survivors equal to the restatement's; a frame's logits do not depend on its batch; and the plain throughput handle
    (WTK_NO_SK_MIXED=1, WTK_SMALL_NARROW=0) picks the same survivors."""
    size, B = 384, 15
    depth, width, maxch = ys.SCALES["s"]
    w = ys.synthetic_weights("s", 1, seed=0)
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    mk = lambda: hip.HipYolo(w, (size, size), 16, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan="throughput")
    monkeypatch.setenv("WTK_NO_SK_MIXED", "1")
    monkeypatch.setenv("WTK_SMALL_NARROW", "0")
    plain = mk()
    monkeypatch.delenv("WTK_NO_SK_MIXED")
    monkeypatch.delenv("WTK_SMALL_NARROW")
    mixed = mk()
    assert mixed.plan == plain.plan == "throughput"
    frames = fr.diverse_frames(16, size, seed=808)[:B]
    box_o, cls_o, (xywh_o, conf_o, anchor_o) = _oracle(oracle, frames, size)
    xm, cm, am = mixed.predict_host(frames, conf=0.1)
    bm, km = mixed.debug_head(B)
    np.testing.assert_allclose(km, cls_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_allclose(bm, box_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_array_equal(am, anchor_o)
    np.testing.assert_allclose(xm, xywh_o, rtol=0, atol=BOX_ATOL)
    xp, cp, ap = plain.predict_host(frames, conf=0.1)
    bp, kp = plain.debug_head(B)
    np.testing.assert_array_equal(ap, am)
    assert not np.array_equal(kp, km)  # the two really are different kernels on some layers (K summed in another order)
    np.testing.assert_allclose(kp, km, rtol=1e-3, atol=LOGIT_ATOL)
    x1, c1, a1 = mixed.predict_host(frames[6:7], conf=0.1)
    b1, k1 = mixed.debug_head(1)
    np.testing.assert_array_equal(k1, km[6:7])
    np.testing.assert_array_equal(b1, bm[6:7])
    plain.close(), mixed.close()


@pytest.mark.parametrize("size,B", [(384, 15), (640, 7), (128, 3)])
def test_six_slab_window_ring_is_bit_identical_to_the_three_slab_kernel(hip_lib, size, B, monkeypatch):
    """A small f16x3 handle runs its 64-cout x 128-pixel window tiles on the six-slab ring with fragment prefetch (conv3x3_halo.hip, NWB = 6): the same
    taps in the same order, so raw head tensors, survivors and boxes equal the three-slab kernel's (WTK_HALO_DEEP=0) bit for bit — for a full call and
    for one frame of it."""
    depth, width, maxch = ys.SCALES["s"]
    w = ys.synthetic_weights("s", 1, seed=2)
    mk = lambda: hip.HipYolo(w, (size, size), 16, dtype="f16x3", width=width, depth=depth, max_channels=maxch, plan="throughput")
    for v in ("WTK_NO_SK_MIXED", "WTK_SMALL_NARROW", "WTK_HALO_DEEP"):
        monkeypatch.delenv(v, raising=False)  # the rules as a user gets them: narrow tiles are what the six-slab ring serves
    deep = mk()
    monkeypatch.setenv("WTK_HALO_DEEP", "0")
    three = mk()
    frames = fr.diverse_frames(16, size, seed=99)[:B]
    xd, cd, ad = deep.predict_host(frames, conf=0.1)
    bd, kd = deep.debug_head(B)
    xt, ct, at = three.predict_host(frames, conf=0.1)
    bt, kt = three.debug_head(B)
    for got, want in ((xd, xt), (cd, ct), (ad, at), (bd, bt), (kd, kt)):
        np.testing.assert_array_equal(got, want)
    x1, c1, a1 = deep.predict_host(frames[B - 1:B], conf=0.1)
    b1, k1 = deep.debug_head(1)
    np.testing.assert_array_equal(k1, kd[B - 1:B])
    np.testing.assert_array_equal(b1, bd[B - 1:B])
    deep.close(), three.close()


@pytest.mark.parametrize("dtype", ["f16x3", "fp32"])
def test_latency_plan_with_a_stock_80_class_head(hip_lib, dtype):
    """nc = 80 (a stock YOLOv8 head): the class towers' last 1x1 stores 80 of 96 padded couts, the class logits are [A, 80] fp32 rows."""
    size, B, nc = 128, 2, 80
    _, det, oracle = _handle(size, dtype, max_batch=4, nc=nc)
    frames = fr.diverse_frames(4, size, seed=17)[:B]
    box_o, cls_o, (xywh_o, conf_o, anchor_o) = _oracle(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    assert cls_g.shape == cls_o.shape == (B, det.anchors, nc)
    np.testing.assert_allclose(cls_g, cls_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=BOX_ATOL)
    det.close()


def test_latency_plan_at_1280(hip_lib):
    """BASELINE config 5's frame size on a latency-plan handle (one frame): 160 x 160 / 80 x 80 / 40 x 40 maps, the largest buffers the kernel's 32-bit
    lane offsets see."""
    size, B = 1280, 1
    _, det, oracle = _handle(size, "f16x3", max_batch=2)
    frames = fr.diverse_frames(4, size, seed=1280)[:B]
    box_o, cls_o, (xywh_o, conf_o, anchor_o) = _oracle(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    np.testing.assert_allclose(cls_g, cls_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o, rtol=1e-3, atol=LOGIT_ATOL)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=BOX_ATOL)
    det.close()


@pytest.mark.parametrize("plan,n_handles", [("auto", 2), ("latency", 1), ("throughput", 1)])
def test_closed_loop_controller_on_every_plan_equals_the_oracle_controller(hip_lib, tmp_path, monkeypatch, plan, n_handles):
    """The closed loop (camera views depend on the previous cycle's movement) through HipYoloController as a user gets it — none of the suite's plan
    switches set: plan "auto" = the single-frame call on a latency-plan handle and the 9-frame cycle batch on a small throughput-plan handle (split-K
    kernel on its smallest maps, 64-cout tiles on thin grids); "latency" / "throughput" = one handle for both.  Integer platform moves and logged boxes
    must be the CPU-restatement controller's, for host crops and for device-resident frames."""
    from harness.sim_harness import ArrayReader, Simulator
    from oracle.controllers_oracle import OracleYoloController
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger

    for v in ("WTK_LATENCY_PLAN", "WTK_NO_SK_MIXED", "WTK_SMALL_NARROW"):
        monkeypatch.delenv(v, raising=False)
    w = ys.synthetic_weights("s", 1, seed=0)
    path = str(tmp_path / "s.wtk")
    ys.save_weights(path, w, "s", 1)
    depth, width, maxch = ys.SCALES["s"]
    frames, _ = fr.synthetic_frames(40, 256, seed=8)
    ec = ExperimentConfig("synthetic", 40, 60, (256, 256), 32, (128, 128))

    def run(make, deferred=False):
        tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.5, 0.5))
        ctrl = make(tc)
        moves = []
        inner = ctrl.provide_movement_vector

        def wrapped(sim):
            m = inner(sim)
            moves.append((int(m[0]), int(m[1])))
            return m

        ctrl.provide_movement_vector = wrapped
        log = TrackLogger(ctrl, deferred=deferred)
        Simulator(tc, ec, log, reader=ArrayReader(frames)).run()
        return moves, log.rows, ctrl

    cfg = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="f16x3", scale="s", plan=plan)
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    m_g, rows_g, ctrl = run(lambda tc: HipYoloController(tc, cfg))
    plans = sorted(d.plan for d in ctrl._model._dets.values())
    assert len(plans) == n_handles and plans == (["latency", "throughput"] if plan == "auto" else [plan])
    assert all(d.max_batch == (4 if plan == "auto" and d.plan == "latency" else 16) for d in ctrl._model._dets.values())  # (auto: the latency handle is sized for the calls it sees)
    m_o, rows_o, _ = run(lambda tc: OracleYoloController(tc, oracle, imgsz=128, conf=0.1))
    assert m_g == m_o and len(m_g) == 4
    m_d, rows_d, _ = run(lambda tc: HipYoloController(tc, cfg, device_frames=torch.from_numpy(frames).cuda()))
    assert m_d == m_g and rows_d == rows_g
    # the deferred track log: the cycle batch enqueued on the controller's second lane and collected a cycle later (plan "auto": it runs beside the next
    # single-frame call on a handle of its own; one shared handle: the lanes take turns) — same moves, same rows
    m_p, rows_p, ctrl_p = run(lambda tc: HipYoloController(tc, cfg, device_frames=torch.from_numpy(frames).cuda()), deferred=True)
    assert m_p == m_g and rows_p == rows_g
    assert sorted(k[0] for k in ctrl_p._view_bufs) == [0, 1] and not ctrl_p._inflight
    assert len(rows_g) == len(rows_o) == 36
    for a, b in zip(rows_g, rows_o):
        assert (a["frame"], a["cycle"], a["phase"], a["plt_x"], a["plt_y"]) == (b["frame"], b["cycle"], b["phase"], b["plt_x"], b["plt_y"])
        np.testing.assert_allclose([a["wrm_x"], a["wrm_y"], a["wrm_w"], a["wrm_h"]], [b["wrm_x"], b["wrm_y"], b["wrm_w"], b["wrm_h"]], atol=BOX_ATOL)


@pytest.mark.parametrize("dtype,tol", [("f16x3", 2e-5), ("fp32", 2e-5)])
def test_every_conv_tensor_of_the_latency_plan_against_the_large_batch_kernels(hip_lib, dtype, tol, monkeypatch):
    """Layer by layer: the output tensor of every conv blob (wtk_yolo_debug_tensor) on a latency-plan handle against the same model on the large-batch
    kernels (Detect tails unfused there so that their inputs exist as tensors): 3x3 / 1x1, stride 1 / 2, residual, two-source upsample loader, fp32
    Detect outputs — every variant conv_sk_kernel serves, each within 2e-5 of the tensor's scale (K is summed in another order, nothing else differs)."""
    size, B = 384, 3
    monkeypatch.setenv("WTK_NO_FUSED_TAIL", "1")
    monkeypatch.setenv("WTK_FRONT_DEBUG", "1")
    monkeypatch.setenv("WTK_NO_SK_MIXED", "1")
    monkeypatch.setenv("WTK_SMALL_NARROW", "0")
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    frames = fr.diverse_frames(4, size, seed=4242)[:B]
    outs = {}
    for plan in ("throughput", "latency"):
        det = hip.HipYolo(w, (size, size), 4, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan=plan)
        det.predict_host(frames, conf=0.1)
        tensors = {}
        for i, t in enumerate(ys.conv_table("s", 1)):
            try:
                tensors[t["name"]] = det.debug_tensor(i, B)
            except hip.WtkError:
                pass  # a conv computed inside another op (concatenated Detect stems report under their first blob)
        outs[plan] = tensors
        det.close()
    a, b = outs["throughput"], outs["latency"]
    assert set(a) == set(b) and len(a) >= 50
    worst = ("", 0.0)
    for nm in a:
        scale = max(1.0, float(np.abs(a[nm]).max()))
        err = float(np.abs(a[nm] - b[nm]).max()) / scale
        if err > worst[1]:
            worst = (nm, err)
        assert err < tol, (nm, err, scale)
    print(f"\nlatency plan vs large-batch kernels, {dtype}: worst conv tensor {worst[0]} differs by {worst[1]:.2e} of its scale ({len(a)} tensors)")


@pytest.mark.parametrize("seed", [1, 2, 5])
def test_latency_plan_on_other_weight_draws(hip_lib, seed):
    """Weight draws the plan was never tuned on (gain tables exist for seeds 0-7): logits 2e-3, survivors equal to the restatement's, in calls of 15 and of 1."""
    size, n = 384, 30
    _, det, oracle = _handle(size, "f16x3", seed=seed)
    frames = fr.diverse_frames(32, size, seed=100 + seed)[:n]
    for i in range(0, n, 15):
        box_o, cls_o, (xywh_o, conf_o, anchor_o) = _oracle(oracle, frames[i : i + 15], size)
        xywh, conf, anchor = det.predict_host(frames[i : i + 15], conf=0.1)
        box_g, cls_g = det.debug_head(15)
        np.testing.assert_allclose(cls_g, cls_o, rtol=1e-3, atol=LOGIT_ATOL)
        np.testing.assert_allclose(box_g, box_o, rtol=1e-3, atol=LOGIT_ATOL)
        np.testing.assert_array_equal(anchor, anchor_o)
        np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=BOX_ATOL)
        x1, _, a1 = det.predict_host(frames[i + 3 : i + 4], conf=0.1)
        assert a1[0] == anchor[3]
        np.testing.assert_array_equal(x1[0], xywh[3])
    det.close()
