"""Pins the oracle (oracle/centerface_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tools/gen_goldens.py importing /root/reference)."""
import numpy as np
import pytest
import torch

import centerface_amd as cfa
from oracle import centerface_oracle as O

TOL = dict(rtol=1e-5, atol=2e-5)   # fp32 CPU vs fp32 CPU, same torch ops: only op-fusion level noise


def _sub(d, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def test_synthetic_weights_are_reproducible(golden):
    g = golden("net")
    sd = cfa.weights.synthetic_state_dict(0)
    assert cfa.weights.fingerprint(sd) == str(g["weights_fingerprint"])
    assert list(sd) == list(cfa.schema.state_dict_schema())
    assert sum(int(np.prod(v.shape)) for v in sd.values()) == 1308126


def test_network_heads_match_reference(golden):
    g = golden("net")
    sd = O.to_torch_sd(cfa.weights.synthetic_state_dict(0))
    for tag in "abc":
        out = O.forward(sd, torch.from_numpy(g["x_" + tag]))
        for h in ("hm", "wh", "lm", "reg"):
            np.testing.assert_allclose(out[h].numpy(), g["%s_%s" % (h, tag)], **TOL, err_msg=h + tag)


def test_preprocess_and_sigmoid_clamp(golden):
    g = golden("net")
    sd = O.to_torch_sd(cfa.weights.synthetic_state_dict(0))
    out = O.forward(sd, torch.from_numpy(O.preprocess(g["img_u8"])))
    np.testing.assert_allclose(O.sigmoid_clamp(out["hm"]).numpy(), g["img_hm_sigmoid"], rtol=1e-5, atol=1e-6)
    for h in ("wh", "lm", "reg"):
        np.testing.assert_allclose(out[h].numpy(), g["img_" + h], **TOL)


def test_mbconv_blocks(golden):
    g = golden("ops")
    i = 0
    while "mb%d_cfg" % i in g:
        cin, cout, t, k, s = (int(v) for v in g["mb%d_cfg" % i])
        sd = _sub(g, "mb%d_w_" % i)
        sd = {"blk." + k_: v for k_, v in sd.items()}
        y = O.mbconv(torch.from_numpy(g["mb%d_x" % i]), sd, "blk", cin, cout, t, k, s)
        np.testing.assert_allclose(y.numpy(), g["mb%d_y" % i], **TOL, err_msg="mb%d" % i)
        i += 1
    assert i == 9


def test_conv_swish_flavours(golden):
    g = golden("ops")
    for i in range(6):
        cin, cout, k, s, groups = (int(v) for v in g["cr%d_cfg" % i])
        y = O.conv_swish(torch.from_numpy(g["cr%d_x" % i]), torch.from_numpy(g["cr%d_w" % i]), k, s, groups)
        np.testing.assert_allclose(y.numpy(), g["cr%d_y" % i], **TOL, err_msg="cr%d" % i)


def test_conv_1x1_bn_idaup_head(golden):
    g = golden("ops")
    y = O.conv_1x1_bn(torch.from_numpy(g["c1bn_x"]), _sub(g, "c1bn_w_"))
    np.testing.assert_allclose(y.numpy(), g["c1bn_y"], **TOL)
    for i in range(3):
        y = O.idaup(torch.from_numpy(g["ida%d_lo" % i]), torch.from_numpy(g["ida%d_skip" % i]),
                    _sub(g, "ida%d_w_" % i), "up")
        np.testing.assert_allclose(y.numpy(), g["ida%d_y" % i], **TOL)
    sd = O.to_torch_sd(cfa.weights.synthetic_state_dict(0))
    np.testing.assert_allclose(O.head(torch.from_numpy(g["head_x"]), sd, "lm").numpy(), g["head_y"], **TOL)
    # fill_up_weights on a 2x2 kernel is [[1,0],[0,0]]
    assert np.array_equal(g["fill_up_2x2"][0, 0], np.array([[1, 0], [0, 0]], np.float32))


def test_shufflev2_blocks(golden):
    g = golden("ops")
    for i in range(4):
        inp, oup, mid, k, s = (int(v) for v in g["sh%d_cfg" % i])
        y = O.shuffle_v2_block(torch.from_numpy(g["sh%d_x" % i]), _sub(g, "sh%d_w_" % i), inp, oup, mid, k, s)
        np.testing.assert_allclose(y.numpy(), g["sh%d_y" % i], **TOL, err_msg="sh%d" % i)


def test_decode_d3_bit_exact(golden):
    g = golden("decode_d3")
    for tag in "smlx":
        heat, wh, reg, K = g[tag + "_heat"], g[tag + "_wh"], g[tag + "_reg"], int(g[tag + "_K"])
        assert np.array_equal(O.peak_nms(heat), g[tag + "_nms"])
        sc, inds, cls, ys, xs = O.topk(O.peak_nms(heat), K)
        # scores are strictly distinct above the zero plateau: exact wherever the score is > 0
        pos = g[tag + "_topk_score"] > 0
        assert pos.sum() > 0
        assert np.array_equal(sc, g[tag + "_topk_score"])
        assert np.array_equal(inds[pos], g[tag + "_topk_inds"][pos])
        assert np.array_equal(ys[pos], g[tag + "_topk_ys"][pos]) and np.array_equal(xs[pos], g[tag + "_topk_xs"][pos])
        assert cls.dtype == np.int32 and not cls.any()
        det, _, _ = O.ctdet_decode(heat, wh, reg, K)
        assert np.array_equal(det[pos], g[tag + "_det"][pos])          # bit-exact float32
        det, _, _ = O.ctdet_decode(heat, wh, None, K)
        assert np.array_equal(det[pos], g[tag + "_det_noreg"][pos])
    assert np.array_equal(O.peak_nms(g["tie_heat"]), g["tie_nms"])
    assert g["tie_nms"][0, 0, 2, 2] == g["tie_nms"][0, 0, 2, 3] == np.float32(0.7)   # plateau kept


def test_decode_d1_and_nms(golden):
    g = golden("decode_d1")
    for tag in "ab":
        b, l = O.decode_d1(g[tag + "_hm"], g[tag + "_wh"], g[tag + "_off"], g[tag + "_lm"],
                           tuple(int(v) for v in g[tag + "_size"]), threshold=0.77)
        assert np.array_equal(b, g[tag + "_boxes"]), tag
        assert np.array_equal(l, g[tag + "_lms"]), tag
    assert len(g["a_boxes"]) == 5 and len(g["b_boxes"]) > 20
    b, l = O.decode_d1(np.full((1, 1, 8, 8), 0.2, np.float32), np.ones((1, 2, 8, 8), np.float32),
                       np.zeros((1, 2, 8, 8), np.float32), np.zeros((1, 10, 8, 8), np.float32), (32, 32))
    assert b == [] and l == [] and g["c_empty_is_list"].all()
    for thr in (0.3, 0.5):
        keep = O.nms_greedy(g["nms_boxes"], g["nms_scores"], thr)
        assert np.array_equal(np.asarray(keep), g["nms_keep_%d" % int(thr * 10)])


def test_transform_and_rescale(golden):
    g = golden("decode_d1")
    for (h, w), ref in zip(g["tf_in"], g["tf_out"]):
        assert np.array_equal(np.asarray(O.transform(int(h), int(w)), np.float64), ref)
    dets, lms = O.rescale(g["a_boxes"], g["a_lms"], g["f_scale"][0], g["f_scale"][1])
    assert np.array_equal(dets, g["f_dets"]) and np.array_equal(lms, g["f_lms"])
    d, l = O.rescale([], [], 1.0, 1.0)
    assert d.shape == (0, 5) and l.shape == (0, 10) and d.dtype == np.float32


def test_post_process_affine_analytic():
    """ctdet_post_process / transform_preds (utils/post_process.py:83-100, utils/image.py:19-66): cv2 is
    not installable here, so the restatement is pinned analytically -- for rot = 0 the inverse map is
    x_src = cx + (x - w/2) * (s/w), y_src = cy + (y - h/2) * (s/w)."""
    rng = np.random.default_rng(4)
    for (cx, cy, sc, w, h) in ((320.0, 240.0, 640.0, 160, 120), (100.5, 77.25, 512.0, 128, 128), (360.0, 239.0, 736.0, 184, 120)):
        pts = rng.uniform(0, w, (20, 2)).astype(np.float32)
        out = O.transform_preds(pts, np.array([cx, cy], np.float32), sc, (w, h))
        r = sc / w
        np.testing.assert_allclose(out[:, 0], cx + (pts[:, 0].astype(np.float64) - w / 2) * r, rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(out[:, 1], cy + (pts[:, 1].astype(np.float64) - h / 2) * r, rtol=1e-6, atol=1e-4)
        fwd = O.get_affine_transform(np.array([cx, cy], np.float32), sc, 0, (w, h))
        inv = O.get_affine_transform(np.array([cx, cy], np.float32), sc, 0, (w, h), inv=1)
        eye = np.vstack([fwd, [0, 0, 1]]) @ np.vstack([inv, [0, 0, 1]])
        np.testing.assert_allclose(eye, np.eye(3), atol=1e-9)
    dets = np.zeros((2, 3, 6), np.float32)
    dets[:, :, :4] = rng.uniform(0, 100, (2, 3, 4)); dets[:, :, 4] = [[0.9, 0.8, 0.7]] * 2
    ret = O.ctdet_post_process(dets.copy(), np.array([[50, 50], [60, 40]], np.float32), np.array([200.0, 100.0], np.float32), 100, 100, 1)
    assert len(ret) == 2 and list(ret[0]) == [1] and len(ret[0][1]) == 3 and len(ret[0][1][0]) == 5


def test_decode_d2_vs_reference(golden):
    """eval_widerface.decode (eval_widerface.py:92-152): threshold honoured, offsets swapped + 0.5."""
    g = golden("decode_d2")
    for tag in "abc":
        h, w = g[tag + "_hm"].shape[1:]
        b = O.decode_d2(g[tag + "_hm"], g[tag + "_wh"], g[tag + "_off"], (h * 4, w * 4), threshold=float(g[tag + "_thr"]))
        assert np.array_equal(np.asarray(b, np.float32).reshape(-1, 5), g[tag + "_boxes"].reshape(-1, 5)), tag
    assert len(g["a_boxes"]) > 10 and bool(g["empty_is_list"])
    assert O.decode_d2(np.full((1, 4, 4), 0.1, np.float32), np.ones((2, 4, 4), np.float32),
                       np.zeros((2, 4, 4), np.float32), (16, 16), threshold=0.5) == []


def _eval_case(g):
    """The evaluate fixture as lists of batches (detections as get_detections returns them: [] for an image without boxes)."""
    picked, annots = [], []
    for bi, nimg in enumerate(g["eval_batches"]):
        picked.append([g["eval_b%d_i%d_det" % (bi, j)] if len(g["eval_b%d_i%d_det" % (bi, j)]) else [] for j in range(int(nimg))])
        annots.append([g["eval_b%d_i%d_gt" % (bi, j)] for j in range(int(nimg))])
    return picked, annots


def test_bbox_overlap_and_evaluate_vs_reference(golden):
    """eval_widerface.bbox_overlap (:48-74) bit-exact (float64 matrix of float32 quotients) and evaluate (:172-211) to the last
    bit of its float64 accumulation, incl. touching / disjoint / nested boxes, padding rows and the three empty cases."""
    g = golden("eval_metrics")
    for i in g["ov_cases"]:
        ov = O.bbox_overlap(g["ov%d_boxes" % i], g["ov%d_query" % i])
        assert ov.dtype == np.float64 and np.array_equal(ov, g["ov%d_out" % i]), i
    assert (g["ov2_out"][0, 0] == 1.0) and (g["ov2_out"][1, 1] == 0.0) and (0 < g["ov2_out"][2, 2] < 0.2)
    picked, annots = _eval_case(g)
    for thr, key in ((0.5, "eval_thr50"), (0.35, "eval_thr35")):
        r, p = O.evaluate(picked, annots, threshold=thr)
        assert (r, p) == tuple(g[key]), (thr, r, p, g[key])
    assert O.bbox_overlap(np.zeros((0, 4), np.float32), g["ov0_query"]).shape == (0, len(g["ov0_query"]))


# ----------------------------------------------------------------------------- N4: training-side pieces
def test_target_encoding_and_losses_match_reference(golden):
    """oracle.encode_targets / gaussian_radius / ctdet_loss against outputs of the reference's
    utils/image.py primitives + model/losses.py CtdetLoss (tools/gen_goldens_train.py)."""
    import torch
    g = golden("train")
    for s, r in zip(g["radius_sizes"], g["radius"]):
        assert O.gaussian_radius((int(s[0]), int(s[1]))) == r
    H, W = g["b0_hm"].shape[1:]
    M = g["b0_wh"].shape[0]
    enc = []
    for b in range(3):
        n = int(g["b%d_n" % b])
        t = O.encode_targets(g["b%d_boxes" % b][:n], g["b%d_lms" % b][:n], H, W, M)
        for k in ("hm", "wh", "reg", "ind", "reg_mask", "landmarks", "lm_ind", "lm_mask"):
            assert np.array_equal(t[k], g["b%d_%s" % (b, k)]), (b, k)
        enc.append(t)
    heads = {k: g["heads_" + k] for k in ("hm", "wh", "reg", "lm")}
    for name in ("all", "empty"):
        idx = list(g["loss_%s_idx" % name])
        out = {k: torch.from_numpy(v[idx].copy()) for k, v in heads.items()}
        st = lambda key: torch.from_numpy(np.stack([enc[i][key] for i in idx]))
        batch = {"hm": st("hm"), "reg_mask": st("reg_mask"), "ind": st("ind"), "wh": st("wh"), "reg": st("reg"),
                 "lm_mask": st("lm_mask"), "lm_ind": st("lm_ind"), "lm": st("landmarks")}
        np.testing.assert_allclose(O.ctdet_loss(out, batch), g["loss_" + name], rtol=1e-6, atol=1e-7)


def _getitem_inputs(g, i):
    """Raw WIDER-style annotation rows of sample i -> (boxes x1y1x2y2, landmarks or -1) as dataset.py:96-110 builds them."""
    anns = g["s%d_anns" % i]
    boxes = np.array([[a[0], a[1], a[0] + a[2], a[1] + a[3]] for a in anns], np.float32)
    lms = -np.ones((len(anns), 10), np.float32)
    for k, a in enumerate(anns):
        if a[4] >= 0:
            for j in range(5):
                lms[k, 2 * j], lms[k, 2 * j + 1] = a[4 + 3 * j], a[5 + 3 * j]
    return boxes, lms


def test_target_encoding_matches_reference_getitem(golden):
    """The oracle's dataset_to_output_map + encode_targets against the dict the reference's OWN
    ``CenterFaceData.__getitem__`` returned (tools/gen_goldens_getitem.py drives the class itself, split = "train", seeded
    random scale / centre / flip; tests/golden/train_getitem.npz) -- every target tensor bit-exact."""
    g = golden("train_getitem")
    total = 0
    for i in range(int(g["n_samples"])):
        h, w = (int(v) for v in g["s%d_size" % i])
        boxes, lms = _getitem_inputs(g, i)
        ob, ol = O.dataset_to_output_map(boxes[:128], lms[:128], g["s%d_c" % i], float(g["s%d_s" % i]), 160, 160,
                                         flipped=bool(g["s%d_flipped" % i]), width=w)
        t = O.encode_targets(ob, ol, 160, 160, 128)
        for k, gk in (("hm", "hm"), ("wh", "wh"), ("reg", "reg"), ("ind", "ind"), ("reg_mask", "reg_mask"),
                      ("landmarks", "lm"), ("lm_ind", "lm_ind"), ("lm_mask", "lm_mask")):
            assert np.array_equal(t[k], g["s%d_%s" % (i, gk)]), (i, k)
        total += int(t["reg_mask"].sum())
    assert total > 50


def test_reference_format_checkpoint_round_trip(tmp_path):
    """A checkpoint written the way the reference writes it (train.py:165: torch.save(model.state_dict()), an
    OrderedDict of tensors incl. the int64 num_batches_tracked buffers) loads through weights.load_checkpoint
    into the exact arrays; a missing / extra / mis-shaped tensor is refused like load_state_dict(strict=True)."""
    import torch
    import centerface_amd as cfa
    from collections import OrderedDict
    sd = cfa.weights.synthetic_state_dict(3)
    path = str(tmp_path / "model_epoch_100.pt")
    torch.save(OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items()), path)
    back = cfa.weights.load_checkpoint(path)
    assert list(back) == list(sd) and cfa.weights.fingerprint(back) == cfa.weights.fingerprint(sd)
    # wrapped form {'state_dict': ...}
    torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "epoch": 100}, path)
    assert cfa.weights.fingerprint(cfa.weights.load_checkpoint(path)) == cfa.weights.fingerprint(sd)
    bad = dict(sd); bad.pop("hm.1.bias")
    with pytest.raises(ValueError):
        cfa.weights.validate_state_dict(bad)
    bad = dict(sd); bad["extra.weight"] = np.zeros(3, np.float32)
    with pytest.raises(ValueError):
        cfa.weights.validate_state_dict(bad)
    bad = dict(sd); bad["first_conv.0.1.weight"] = np.zeros((32, 3, 5, 5), np.float32)
    with pytest.raises(ValueError):
        cfa.weights.validate_state_dict(bad)


def test_bf16_emulation_matches_hooked_reference(golden):
    """oracle/bf16_emulation.py (the engine's pre-scaled bf16/fp16 arithmetic restated on CPU) against the
    REFERENCE's module graph run with the same quantised weights and rounding hooks (tools/gen_goldens_bf16emu.py):
    end to end on the 32x32 case (no rounding flip occurs: agreement to fp32 noise), and block by block on the
    golden's own block inputs for the larger cases, within one bf16 ulp + flip noise."""
    from oracle import bf16_emulation as E
    g = golden("net_bf16emu")
    sd = cfa.weights.synthetic_state_dict(0)
    assert str(g["weights_fingerprint"]) == cfa.weights.fingerprint(sd)
    out = E.forward(sd, x=g["x_a"])
    for h in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(out[h].numpy(), g[h + "_a"], rtol=0, atol=1e-3, err_msg=h)
    for tag in "abc":
        stats = E.check_blockwise(sd, {k[:-2]: v for k, v in g.items() if k.endswith("_" + tag)}, detail=True)
        assert len(stats) == 20, sorted(stats)
        # small maps (down to 320 elements): a single flipped element is already 3e-3 of a tensor, so the fraction
        # criteria of the GPU tests (E.accept) are replaced by "few beyond the bound, none beyond twice the bound"
        bad = {k: v for k, v in stats.items() if v[0] > 2.0 or v[1] * v[4] > max(2, 1e-4 * v[4])}
        assert not bad, (tag, bad)
    # the uint8 staging (one fma per byte) end to end: drift-level agreement only (see the emulation's docstring)
    out = E.forward(sd, img_u8=g["img_u8"])
    for h in ("hm", "wh", "lm", "reg"):
        ref = g["img_" + h]
        d = np.abs(out[h].numpy() - ref)
        assert d.mean() < 0.006 * np.sqrt((ref ** 2).mean()), h


def test_f16_round_toward_zero_helper():
    from oracle import bf16_emulation as E
    t = torch.tensor([1.0009765625 + 1e-4, -1.0009765625 - 1e-4, 3e-6, -3e-6, 7e4, -7e4, 65503.9, 1e-8, 0.0, 2.0 ** -14, 2.0 ** -14 - 1e-9])
    r = E.q_f16_rtz_sat(t)
    exp = torch.tensor([1.0009765625, -1.0009765625, 2.0 ** -24 * 50, -(2.0 ** -24) * 50, 65504.0, -65504.0, 65472.0, 0.0, 0.0, 2.0 ** -14,
                        2.0 ** -14 - 2.0 ** -24])
    assert torch.equal(r, exp), (r, exp)
    # round-to-nearest saturating (tap packer)
    assert torch.equal(E.q_f16_rne_sat(torch.tensor([7e4, 65519.0, 1.00048828125])), torch.tensor([65504.0, 65504.0, 1.0]))


def test_cv2_fixed_point_resize_known_answers():
    """resize_bilinear_u8 = OpenCV's fixed-point INTER_LINEAR for uint8 (resize.cpp; cv2 itself is not installable
    here).  Known answers derived by hand from the published algorithm: identity, a 2x horizontal up-sample of
    [0, 255] (column coefficients (2048,0), (1536,512), (512,1536), (2048,0) -> r = 0, 130560, 391680, 522240 ->
    ((2048 * (r >> 4)) >> 16) + 2 >> 2 = 0, 64, 191, 255), a 2x vertical one, and monotonic / range properties."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_bilinear_u8(img, 37, 53), img)
    two = np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8)                   # 1 x 2
    assert O.resize_bilinear_u8(two, 1, 4)[0, :, 0].tolist() == [0, 64, 191, 255]
    col = two.transpose(1, 0, 2)                                              # 2 x 1: rows are clamped, coefficients kept
    # dy=0: fy=-0.25 -> sy=-1 (both rows clamp to 0) ; dy=1: (1536, 512) ; dy=2: (512, 1536) ; dy=3: sy=1, rows (1, 1)
    assert O.resize_bilinear_u8(col, 4, 1)[:, 0, 0].tolist() == [0, 64, 191, 255]
    flat = np.full((9, 11, 3), 137, np.uint8)
    assert (O.resize_bilinear_u8(flat, 32, 32) == 137).all()                  # constants survive (coefficients sum to 2048)
    up = O.resize_bilinear_u8(img, 64, 64)
    assert up.min() >= img.min() and up.max() <= img.max()
    # transform() + resize: the geometry of BASELINE configs[0] (imgs/1.jpg is 720 x 478 -> 736 x 480)
    assert O.transform(478, 720)[:2] == (480, 736)
