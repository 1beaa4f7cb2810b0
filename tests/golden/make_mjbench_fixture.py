#!/usr/bin/env python3
"""Pins the MJ-BENCH-VIDEO evaluation bookkeeping with the reference's OWN methods.

Build-container only (needs /root/reference).  ``CustomTrainer.evaluate`` / ``evaluate_aspect`` / ``calculate_metrics`` /
``save_metrics`` (scripts/train/overall_train.py:204-442) are compiled out of the reference's file into a bare class (the
module itself imports transformers.Trainer / accelerate / the S3 dataset, none of which the methods need) and EXECUTED on
synthetic pairs: labels in the datas/test.json schema drawn from integer seeds, turned into batches by the reference's own
``process_labels`` / ``deal_preference`` (dataset.py:52-112) laid out as VideoDataCollator lays them out
(dataset.py:406-520), and a stub model that replays synthetic ``CustomOutput`` fields.  What is stored
(tests/golden/mjbench.json): the seeds, and the numbers the reference's methods returned / wrote - nothing of its text.
tests/test_host_fixtures.py holds mj_video_amd.harness.evaluate_mjbench to them on the CPU.

Usage:  python tests/golden/make_mjbench_fixture.py
"""
import ast
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

CRITERIA = ["object", "attribute", "actions", "count", "location", "Crime", "Shocking", "Disgust", "NSFW Evasive", "NSFW Subtle",
            "Political Sensitivity", "Human Face Distortion", "Human Limb Distortion", "Object Distortion", "De-focused Blurred",
            "Motion Blurred", "Spatial Consistency", "Action Continuity", "Object Disappearance", "Abrupt Background Changes",
            "Inconsistent Lighting Shadows", "Frame Flickering", "Object Drift", "Race", "Age", "Education", "Job", "Gender"]
ASPECTS = ["Alignment", "Safety", "Fineness", "Consistency", "Bias"]
PREF_KEYS = ["Alignment", "Safety", "Fineness", "Coherence & Consistency", "Bias"]


def synth_items(seed: int, n: int):
    """``n`` pairs in the datas/test.json schema (labels only), from numpy's Philox stream ``seed``"""
    g = np.random.Generator(np.random.Philox(key=[seed, 77]))
    items = []
    for i in range(n):
        it = {"caption": f"synthetic caption {seed}-{i}", "video_0_path": f"test/{i}_0.mp4", "video_1_path": f"test/{i}_1.mp4"}
        for v in (0, 1):
            it[f"video_{v}_label"] = {k: int(g.choice([0, 1, 2], p=[0.5, 0.35, 0.15])) for k in CRITERIA}
            it[f"video_{v}_overall_score"] = {k: int(g.choice([0, 1, 2], p=[0.3, 0.45, 0.25])) for k in ASPECTS}
            it[f"video_{v}_total_score"] = int(g.choice([0, 1, 2]))
        it["category_preference"] = {k: str(g.choice(["Video 1 better", "Video 2 better", "Same"])) for k in PREF_KEYS}
        it["overall_preference"] = str(g.choice(["Video 1 better", "Video 2 better", "Same"], p=[0.4, 0.4, 0.2]))
        it["discard"] = False
        items.append(it)
    return items


def synth_scores(seed: int, n: int):
    """[n, 2, 34] fp32 block (score, 5 aspect scores, 28 rewards per video); a few exact zeros and ties on purpose"""
    g = np.random.Generator(np.random.Philox(key=[seed, 78]))
    s = g.standard_normal((n, 2, 34)).astype(np.float32)
    s[g.random((n, 2, 34)) < 0.02] = 0.0
    tie = g.random(n) < 0.05
    s[tie, 1, 0] = s[tie, 0, 0]
    return s


def reference_trainer_class():
    """the four methods, compiled from the reference file into a class without its Trainer base"""
    path = os.path.join(REF, "scripts", "train", "overall_train.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CustomTrainer")
    want = {"evaluate_aspect", "calculate_metrics", "save_metrics", "evaluate"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {f.name for f in fns} == want
    bare = ast.ClassDef(name="RefTrainer", bases=[], keywords=[], body=fns, decorator_list=[])
    mod = ast.Module(body=[bare], type_ignores=[])
    ast.fix_missing_locations(mod)
    written = []

    class _DF:   # save_metrics ends in pd.DataFrame(rows).to_excel(...): keep the rows instead
        def __init__(self, rows):
            self.rows = rows

        def to_excel(self, name, index=False):
            written.append((name, self.rows))

    ns = {"torch": torch, "pd": types.SimpleNamespace(DataFrame=_DF)}
    exec(compile(mod, path, "exec"), ns)
    return ns["RefTrainer"], written


def reference_batches(items, batch_size):
    """VideoDataset.__iter__ + VideoDataCollator.__call__ for the label fields (dataset.py:329-520), with the reference's
    own process_labels / deal_preference"""
    path = os.path.join(REF, "scripts", "data_processor", "dataset.py")
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("process_labels", "deal_preference")]
    ns = {}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    rows = []
    for it in items:
        row = {}
        for v in (0, 1):
            cs, cr, _ = ns["process_labels"](it[f"video_{v}_label"])
            a_s, a_r, _ = ns["process_labels"](it[f"video_{v}_overall_score"])
            row[f"video_{v}_criteria_score"], row[f"video_{v}_criteria_related"] = torch.tensor(cs), torch.tensor(cr)
            row[f"video_{v}_aspect_score"], row[f"video_{v}_aspect_related"] = torch.tensor(a_s), torch.tensor(a_r)
        p, m = ns["deal_preference"](it["overall_preference"], True)
        row["overall_preference"], row["overall_mask"] = torch.tensor(p), torch.tensor(m)
        rows.append(row)
    batches = []
    for lo in range(0, len(rows), batch_size):
        chunk = rows[lo:lo + batch_size]
        b = {k: torch.stack([r[k] for r in chunk]) for k in chunk[0]}
        for v in (0, 1):   # shapes only: the methods read batch sizes off the pixel tensor and pass the rest to the model
            b[f"video_{v}_pixel_values"] = torch.zeros(len(chunk), 2, 3, 4, 4)
            b[f"video_{v}_input_ids"] = torch.zeros(len(chunk), 4, dtype=torch.long)
            b[f"video_{v}_attention_mask"] = torch.ones(len(chunk), 4, dtype=torch.long)
        batches.append(b)
    return batches


class ReplayModel:
    """stands where the reward model stands in the reference's loops: call k returns the synthetic outputs of the k-th
    (batch, video) in the order the methods call it (video 0 then video 1 of every batch)"""

    def __init__(self, scores, batch_size):
        self.scores, self.bs, self.k = torch.from_numpy(scores), batch_size, 0

    def eval(self):
        return self

    def rewind(self):
        self.k = 0

    def __call__(self, pixel_values, input_ids, attention_mask):
        b, v = divmod(self.k, 2)
        self.k += 1
        blk = self.scores[b * self.bs:(b + 1) * self.bs, v]
        return types.SimpleNamespace(score=blk[:, 0].clone(), aspect_scores=blk[:, 1:6].clone(), rewards=blk[:, 6:].clone())


def clean(x):
    if isinstance(x, float) and math.isnan(x):
        return "nan"
    return x


def main():
    Ref, written = reference_trainer_class()
    cases = []
    for seed, n, bs in ((1, 64, 1), (2, 96, 4), (3, 16, 16)):   # (the reference's accumulators need equal batches)
        items, scores = synth_items(seed, n), synth_scores(seed, n)
        batches = reference_batches(items, bs)
        model = ReplayModel(scores, bs)
        tr = Ref()
        tr.model, tr.eval_dataset = model, None
        tr.get_eval_dataloader = lambda ds, _b=batches: _b
        del written[:]
        orig_aspect = Ref.evaluate_aspect

        def aspect_then_rewind(self, *a, **k):   # evaluate() runs evaluate_aspect() first, then its own loop over the same loader
            r = orig_aspect(self, *a, **k)
            model.rewind()
            return r

        tr.evaluate_aspect = types.MethodType(aspect_then_rewind, tr)
        out = tr.evaluate()
        files = {name.split("_evaluation")[0]: rows for name, rows in written}
        assert set(files) == {"aspect", "criteria"}
        rec = dict(seed=seed, n_pairs=n, batch_size=bs, overall_accuracy=out["eval_accuracy"])
        for kind, rows in files.items():
            m = {r["Metric"]: float(r["Value"]) for r in rows}
            # the reference flattens [batch, dims] and keeps one counter per (position in the batch, dim): at batch size 1
            # its "dim" is the label dimension; at larger batches the counters are folded over the batch position here
            ndim = 5 if kind == "aspect" else 28
            fold = {c: [sum(m[f"{c} (dim {b * ndim + d})"] for b in range(bs)) for d in range(ndim)] for c in ("TP", "FP", "TN", "FN")}
            rec[kind] = {k: clean(m[k]) for k in ("Accuracy", "Precision", "Recall", "F1 Score", "TP Sum", "FP Sum", "TN Sum", "FN Sum")}
            rec[kind].update(fold)
            if bs == 1:
                for c in ("Accuracy", "Precision", "Recall", "F1 Score"):
                    rec[kind][f"{c} per dim"] = [clean(m[f"{c} (dim {d})"]) for d in range(ndim)]
        cases.append(rec)
        print(f"seed {seed}: {n} pairs, overall accuracy {out['eval_accuracy']:.4f}, aspect accuracy {rec['aspect']['Accuracy']:.4f}, "
              f"criteria accuracy {rec['criteria']['Accuracy']:.4f}")
    json.dump(dict(generator="tests/golden/make_mjbench_fixture.py", reference="scripts/train/overall_train.py:204-442 executed",
                   cases=cases), open(os.path.join(HERE, "mjbench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
