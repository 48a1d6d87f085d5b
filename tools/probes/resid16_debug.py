#!/usr/bin/env python3
"""Debug: bf16 engine, one polishing step with the fp32 residual stream (resid16=0) and with fp16 rows (resid16=1), several batch sizes."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from conzic_amd import harness, native, synth
from conzic_amd.engine import Engine

L, K, SEED = 10, int(os.environ.get("K", "200")), 4
su = harness.build_synthetic(False, native.PREC_BF16, regular_only=True)
eng = su.engine
hp = Engine.hyper(0.02, 2.0, 0.1)
rng = np.random.default_rng(3)
for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "2,8,16,64").split(",")]:
    emb = rng.standard_normal((B, 512)).astype(np.float32)
    eng.set_image_embeds(emb)
    inp0 = np.array([su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)] * B, dtype=np.int32)
    regular = np.nonzero(su.token_mask[0] > 0)[0]
    inp0[:, SEED:SEED + L] = rng.choice(regular, size=(B, L))
    out = {}
    for r16 in (0, 1):
        eng.set_option("resid16", r16)
        eng.profile_reset()
        try:
            res = eng.step(inp0.copy(), SEED + 3, K, hp, want=("final_score", "clip_ref", "best"))
        except Exception as ex:
            print(f"B={B} resid16={r16}: {ex}")
            res = None
        out[r16] = res
        print(f"B={B} resid16={r16} rows={eng.stats()['clip_rows']}", flush=True)
    if out[0] and out[1]:
        d = np.abs(out[0]["clip_ref"] - out[1]["clip_ref"])
        print(f"B={B}: max |d cos| {d.max():.3e} nan={np.isnan(out[1]['clip_ref']).sum()} max |d final| {np.abs(out[0]['final_score'] - out[1]['final_score']).max():.3e} "
              f"winners same {(out[0]['best'] == out[1]['best']).mean():.3f}", flush=True)
        if np.isnan(out[1]["clip_ref"]).any():
            bad = np.argwhere(np.isnan(out[1]["clip_ref"]))
            print("  nan at (image, cand):", bad[:10].tolist(), "count per image", np.isnan(out[1]["clip_ref"]).sum(axis=1).tolist())
sys.stdout.flush(); os._exit(0)
