#!/usr/bin/env python3
"""MJ-BENCH-VIDEO evaluation of the MJ-VIDEO reward model on the MI355X path.

Counterpart of the reference's evaluation of ``datas/test.json`` inside its trainer (scripts/train/overall_train.py:204-442,
``CustomTrainer.evaluate`` -> ``evaluate_aspect``): for every pair, the same caption is scored against both videos
(``num_segments`` frames each, "Frame{i}: <image>\\n" prefix, dataset.py:357-366) and the run reports
  * the overall preference accuracy (prefer_predict = not (score_0 > score_1), counted where ``overall_preference`` is
    decisive - overall_train.py:425-434),
  * the aspect and criteria sign metrics of ``evaluate_aspect`` (accuracy / precision / recall / F1, pooled and per label
    dimension, with the reference's label conventions - ``mj_video_amd.harness.ConfusionCounts``).
The bookkeeping is held to numbers the reference's own methods produced (tests/golden/mjbench.json,
tests/test_host_fixtures.py).  Differences from the reference's loop: both videos of ``--pairs_per_batch`` pairs go through
ONE packed forward, pairs are sharded data-parallel over the GPUs of a node when launched with torch.distributed.run (one
all-gather of the [pairs, 2, 34] score block), frames can be preprocessed on the GPU, videos are read from ``--root`` (no S3).

    python scripts/eval/eval_mjbench_video.py --model_name <dir> --checkpoint_path <dir> --json datas/test.json --root datas/videos
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mj_video_amd  # noqa: E402,F401
from mj_video_amd import harness, parallel  # noqa: E402
from eval_genai_mjvideo import build_model, load_pixels  # noqa: E402  (same model set-up order, same frame loaders)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", required=True, help="local directory with config.json (+ tokenizer files)")
    ap.add_argument("--checkpoint_path", default=None, help="directory holding the MJ-VIDEO *.safetensors checkpoint")
    ap.add_argument("--json", required=True, help="MJ-BENCH-VIDEO pairs in the datas/test.json schema")
    ap.add_argument("--root", default="./datas/videos", help="directory the video_{0,1}_path entries are relative to")
    ap.add_argument("--overall", action="store_true", help="only pairs with a decisive overall preference (VideoDataset(overall=True))")
    ap.add_argument("--num_segments", type=int, default=2, help="frames per video (VideoDataset default: 2; the eval driver uses 8)")
    ap.add_argument("--max_num", type=int, default=1)
    ap.add_argument("--pairs_per_batch", type=int, default=4)
    ap.add_argument("--num_objectives", type=int, default=28)
    ap.add_argument("--num_aspects", type=int, default=5)
    ap.add_argument("--gating_temperature", type=float, default=1.0)
    ap.add_argument("--gating_hidden_dim", type=int, default=1024)
    ap.add_argument("--gating_n_hidden", type=int, default=3)
    ap.add_argument("--host_preprocessing", action="store_true", help="PIL preprocessing on the host instead of the GPU kernel")
    ap.add_argument("--output", default=None, help="write the metrics as JSON here (rank 0)")
    return ap.parse_args(argv)


def evaluate_items(model, config, tokenizer, items, pixel_loader, root, pairs_per_batch=4, generation_config=None):
    """Scores the pairs of ``items`` (data-parallel when a process group exists) and returns (metrics dict, scores
    [n, 2, 34]) - identical on every rank."""
    generation_config = generation_config if generation_config is not None else {"max_new_tokens": 1024, "do_sample": True}

    def score_fn(local):
        blocks = []
        for i in range(0, len(local), pairs_per_batch):
            chunk = [dict(prompt=it["caption"], left_pixels=pixel_loader(os.path.join(root, it["video_0_path"])),
                          right_pixels=pixel_loader(os.path.join(root, it["video_1_path"]))) for it in local[i:i + pairs_per_batch]]
            blocks.append(harness.score_pair_batch(model, config, tokenizer, chunk, generation_config).float())
        return torch.cat(blocks) if blocks else torch.zeros(0, 2, parallel.SCORE_WIDTH, device=model.model.device)

    scores = parallel.score_pairs_dp(score_fn, list(items), device=model.model.device)
    return harness.evaluate_mjbench(items, scores.cpu().numpy()), scores


def main(argv=None):
    args = parse_args(argv)
    from mj_video_amd import _lib
    _lib.assert_product_library()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(args.model_name, trust_remote_code=True)
    config, model = build_model(args, tokenizer, device)
    items = json.load(open(args.json, encoding="utf-8"))
    if args.overall:   # dataset.py:338-339
        items = [it for it in items if it["overall_preference"] in ("Video 1 better", "Video 2 better")]
    metrics, _ = evaluate_items(model, config, tokenizer, items, lambda p: load_pixels(p, args, device), args.root, args.pairs_per_batch)
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(f"Evaluation Results - Accuracy: {metrics['overall_accuracy']:.4f}")   # overall_train.py:440
        for kind in ("aspect", "criteria"):
            m = metrics[kind]
            print(f"{kind}: Accuracy {m['accuracy']:.4f} Precision {m['precision']:.4f} Recall {m['recall']:.4f} F1 {m['f1']:.4f}")
            for name, a, f1 in zip(metrics[f"{kind}_names"], m["accuracy_dim"], m["f1_dim"]):
                print(f"    {name:32s} Acc {a:.4f} F1 {f1:.4f}")
        if args.output:
            json.dump(metrics, open(args.output, "w"), indent=1)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
