"""CPU: the build's host-side code (prompt construction, frame sampling, dynamic tiling, label targets) against
tests/golden/host.json - outputs of the REFERENCE'S OWN functions executed in the build container by
tests/golden/make_host_fixtures.py (prepare_chat_input: internvl2/modeling_internvl_chat.py:36-89; get_index /
dynamic_preprocess: scripts/data_processor/data.py:66-137; process_labels / deal_preference:
scripts/data_processor/dataset.py:52-112)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

from util import GOLDEN, make_cfg
from mj_video_amd import chat_input, harness, video
from test_host_logic import StubTokenizer

HOST = json.load(open(os.path.join(GOLDEN, "host.json")))


def sha(b):
    return hashlib.sha1(b).hexdigest()


@pytest.mark.parametrize("case", HOST["prepare_chat_input"], ids=lambda c: c["name"])
def test_prepare_chat_input_matches_the_reference(case):
    cfg = make_cfg(case["kind"], case["image_size"])
    px = torch.zeros(case["n_tiles"], 3, case["image_size"], case["image_size"])
    gen = {"max_new_tokens": 1024, "do_sample": True}
    hist = [tuple(h) for h in case["history"]] if case["history"] else None
    ids, mask = chat_input.prepare_chat_input(cfg, StubTokenizer(), px, case["question"], gen, history=hist,
                                              num_patches_list=case["num_patches_list"])
    got = ids[0].tolist()
    assert len(got) == case["n_tokens"] and got.count(92546) == case["n_img_context"]
    assert sha(np.asarray(got, np.int64).tobytes()) == case["ids_sha1"]
    if "ids" in case:
        assert got == case["ids"]
    assert bool(mask.all()) == case["mask_all_ones"] and mask.shape == ids.shape
    assert gen["eos_token_id"] == case["eos_token_id"]   # the reference's side effect on the caller's dict (:87)


@pytest.mark.parametrize("case", HOST["get_index"], ids=lambda c: f"{c['bound']}-{c['max_frame']}-{c['num_segments']}")
def test_get_index_matches_the_reference(case):
    b = tuple(case["bound"]) if case["bound"] else None
    got = video.get_index(b, case["fps"], case["max_frame"], first_idx=case["first_idx"], num_segments=case["num_segments"])
    assert [int(x) for x in got] == case["indices"]


def _synthetic_image(w, h, seed):
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[seed, w * 10007 + h]))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], axis=-1)
    return Image.fromarray(((base + g.integers(0, 64, size=(h, w, 3))) % 256).astype(np.uint8), "RGB")


@pytest.mark.parametrize("case", HOST["dynamic_preprocess"], ids=lambda c: f"{c['width']}x{c['height']}-max{c['max_num']}")
def test_dynamic_preprocess_matches_the_reference(case):
    """tile grid choice, crop order, thumbnail rule AND the tiles' pixels (PIL's default bicubic resize of the whole frame)"""
    img = _synthetic_image(case["width"], case["height"], case["image_seed"])
    tiles = video.dynamic_preprocess(img, min_num=case["min_num"], max_num=case["max_num"], image_size=case["image_size"],
                                     use_thumbnail=case["use_thumbnail"])
    assert len(tiles) == case["n_tiles"]
    assert [list(t.size) for t in tiles] == case["tile_sizes"]
    assert [sha(np.asarray(t.convert("RGB")).tobytes()) for t in tiles] == case["tile_sha1"]


def test_label_targets_match_the_reference():
    for rec in HOST["process_labels"]:
        if rec["overall"]:
            s, r = harness.overall_target(rec["labels"], mse=rec["mse"])
            assert (s, r, []) == (rec["score"], rec["related"], rec["names"]), rec
        else:
            assert harness.criteria_targets(rec["labels"], mse=rec["mse"]) == (rec["score"], rec["related"], rec["names"]), rec
    for rec in HOST["deal_preference"]:
        assert harness.preference_targets(rec["labels"]) == (rec["preference"], rec["mask"]), rec


def test_mjbench_video_bookkeeping_matches_the_reference():
    """tests/golden/mjbench.json: numbers the reference's own CustomTrainer.evaluate / evaluate_aspect / calculate_metrics /
    save_metrics (scripts/train/overall_train.py:204-442, executed by make_mjbench_fixture.py) produced on synthetic
    datas/test.json-schema labels and synthetic model outputs, both regenerated here from the stored seeds:
    harness.evaluate_mjbench must give the same overall preference accuracy, the same pooled aspect / criteria accuracy,
    precision, recall, F1 and the same TP / FP / TN / FN per label dimension (and, at batch size 1, where the reference's
    "dim" is the label dimension, the same per-dimension metrics, NaN for dimensions that never occur)."""
    import math
    sys.path.insert(0, GOLDEN)
    import make_mjbench_fixture as gen
    fix = json.load(open(os.path.join(GOLDEN, "mjbench.json")))
    assert fix["cases"]
    for case in fix["cases"]:
        items, scores = gen.synth_items(case["seed"], case["n_pairs"]), gen.synth_scores(case["seed"], case["n_pairs"])
        got = harness.evaluate_mjbench(items, scores)
        assert got["overall_accuracy"] == case["overall_accuracy"], case["seed"]
        for kind in ("aspect", "criteria"):
            ref, mine = case[kind], got[kind]
            for a, b in (("Accuracy", "accuracy"), ("Precision", "precision"), ("Recall", "recall"), ("F1 Score", "f1")):
                assert mine[b] == ref[a], (case["seed"], kind, a, mine[b], ref[a])
            for a, b in (("TP", "tp"), ("FP", "fp"), ("TN", "tn"), ("FN", "fn")):
                assert mine[b] == ref[a], (case["seed"], kind, a)
                assert sum(mine[b]) == ref[f"{a} Sum"]
            if "Accuracy per dim" in ref:
                for a, b in (("Accuracy", "accuracy_dim"), ("Precision", "precision_dim"), ("Recall", "recall_dim"), ("F1 Score", "f1_dim")):
                    for x, y in zip(mine[b], ref[f"{a} per dim"]):
                        if y == "nan":
                            assert math.isnan(x)
                        else:   # the reference computes these ratios in fp32 tensors
                            assert abs(x - y) <= 1e-6 * max(1.0, abs(y)), (case["seed"], kind, a, x, y)
