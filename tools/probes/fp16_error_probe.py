#!/usr/bin/env python3
"""Measurement behind the screen-then-refine engine: the structure of the cosine error of the single-pass
fp16 CLIP-text tower at the published logit scale, on the full-size scale-100 golden (teacher-forced, image embeddings
from the golden so that only the text tower contributes).

For every image-step: d_k = cos_fp16[k] - cos_ref[k]; its mean, its softmax-weighted mean, spread, and what the fused
score error becomes when the candidates with softmax mass above a threshold get the exact cosine instead (simulated
refinement, with and without removing the mean error measured on the refined set from the unrefined ones).

    python tools/probes/fp16_error_probe.py [golden ...]   ->  one JSON line per case
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conzic_amd import harness, native  # noqa: E402
from conzic_amd.engine import Engine  # noqa: E402
from goldutil import load_case  # noqa: E402

SEED_LEN = 4


def softmax(x):
    e = np.exp(x - x.max())
    return e / e.sum()


def run(name, prec):
    meta, arr = load_case(name)
    su = harness.build_synthetic(meta["tiny"], prec, meta["bseed"], meta["cseed"], meta["logit_scale"], meta["regular_only"],
                                 lexicon=meta["gamma"] is not None)
    eng = su.engine
    eng.set_image_embeds(arr["image_embeds"])
    hp = Engine.hyper(meta["alpha"], meta["beta"], meta["temperature"], meta["gamma"], meta["style"] == "negative")
    scale = float(np.exp(meta["logit_scale"]))
    beta = meta["beta"]
    rows = []
    dump = dict(c_ref=[], c_hat=[], probs=[])
    n = arr["probs"].shape[0]
    for i in range(n):
        if meta["reuse"][i]:
            continue
        pos = meta["positions"][i]
        inp = np.ascontiguousarray(arr["inp_before"][i], dtype=np.int32)
        res = eng.step(inp, SEED_LEN + pos, meta["K"], hp, n_mask=1, dot_allowed=(pos == meta["L"] - 1),
                       want=("idxs", "clip_score", "clip_ref", "final_score"))
        B, K = arr["probs"][i].shape
        for b in range(B):
            if not (res["idxs"][b] == arr["idxs"][i][b]).all():
                continue  # top-K set or order differs by a near-tie: skip (rare)
            c_ref = arr["clip_ref"][i][b].astype(np.float64)
            c_hat = res["clip_ref"][b].astype(np.float64)
            d = c_hat - c_ref
            dump["c_ref"].append(arr["clip_ref"][i][b]); dump["c_hat"].append(res["clip_ref"][b]); dump["probs"].append(arr["probs"][i][b])
            p = softmax(scale * c_ref)
            err_plain = beta * np.abs(softmax(scale * c_hat) - p)
            row = dict(step=i, img=b, pmax=float(p.max()), d_mean=float(d.mean()), d_pmean=float((p * d).sum()),
                       d_std=float(d.std()), d_absmax=float(np.abs(d).max()), err_plain=float(err_plain.max()))
            order = np.argsort(-softmax(scale * c_hat))
            ph = softmax(scale * c_hat)
            for theta in (0.02, 0.0125, 0.008, 0.005):
                sel = ph > theta
                for fix in (0, 1):
                    mixed = c_hat.copy()
                    mixed[sel] = c_ref[sel]
                    if fix and sel.any():
                        mixed[~sel] -= d[sel].mean()  # common-mode error estimated on the refined set
                    e = beta * np.abs(softmax(scale * mixed) - p)
                    row[f"n@{theta}"] = int(sel.sum())
                    row[f"err@{theta}/{'fix' if fix else 'raw'}"] = float(e.max())
            rows.append(row)
    eng.close()
    out_dir = os.environ.get("CZC_PROBE_OUT")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        np.savez_compressed(os.path.join(out_dir, f"fp16_cos_{name}.npz"), **{k: np.stack(v) for k, v in dump.items()})
    keys = [k for k in rows[0] if k not in ("step", "img")]
    summ = {k: (float(np.mean([r[k] for r in rows])) if k.startswith("n@") or k in ("pmax",) else float(np.max(np.abs([r[k] for r in rows]))))
            for k in keys}
    summ["d_mean_range"] = [float(min(r["d_mean"] for r in rows)), float(max(r["d_mean"] for r in rows))]
    print(json.dumps(dict(case=name, precision=prec, image_steps=len(rows), worst_or_mean=summ)), flush=True)
    if os.environ.get("CZC_PROBE_ROWS"):
        for r in rows:
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or ["full_scale100"]
    for nm in names:
        run(nm, native.PREC_FP16)
    sys.stdout.flush()
    os._exit(0)
