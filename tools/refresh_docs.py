#!/usr/bin/env python3
"""One source of truth per figure: the numbers README.md, DESIGN.md §0 and the precision block of include/conzic_hip.h quote are
GENERATED from the committed evidence under profiles/ (bench lines, the pytest -m gpu summary, the refine validation), never typed.

    python tools/refresh_docs.py            rewrite the generated blocks in place
    python tools/refresh_docs.py --check    exit 1 when a block is stale (run by the CPU test suite)

A generated block sits between `BEGIN GENERATED <name>` and `END GENERATED <name>` marker lines; everything else in those files is
hand-written prose that may cite a block but carries no measured figure of the current round.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"
P = os.path.join(ROOT, "profiles")


def jload(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    txt = open(path).read().strip()
    try:
        return json.loads(txt.splitlines()[-1])
    except (json.JSONDecodeError, IndexError):
        return None


def jlines(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return []
    out = []
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                out.append(json.loads(ln))
            except json.JSONDecodeError:
                pass
    return out


def facts():
    """Every figure the generated blocks quote, with the file it comes from."""
    f = {}
    drv = jload(f"{ROUND}_bench_driver_cmd.json")
    if drv:
        rf = drv.get("roofline") or {}
        f["value"] = drv["value"]
        f["ms_per_step"] = drv["ms_per_step"]
        f["value_scale100"] = drv.get("value_scale100")
        f["frac"] = rf.get("frac")
        f["achieved"] = rf.get("achieved")
        f["frac_1s"] = (rf.get("single_stream_pass") or {}).get("frac")
        f["tower_util"] = (rf.get("clip_text_mfma_util") or {}).get("frac")
        f["single_stream_value"] = (drv.get("single_stream") or {}).get("value")
        f["traffic"] = rf.get("traffic")
        f["crc"] = (drv.get("captions_crc32") or {}).get("value")
        cb = drv.get("cpu_baseline") or {}
        f["cpu_value"], f["cpu_threads"] = cb.get("value"), cb.get("cores")
        s100 = drv.get("scale100_mode") or {}
        r100 = s100.get("refine") or {}
        f["gated_frac"], f["re_encoded_frac"] = r100.get("gated_frac"), r100.get("re_encoded_frac")
        f["frac_scale100"] = (s100.get("roofline") or {}).get("frac")
        f["executed_tflop"] = drv.get("executed_tflop_per_caption")
        f["calib"] = (drv.get("box_calibration") or {}).get("tflops")
    for tag, name in (("cfg1", "single image"), ("cfg3", "configs[3] shard"), ("cfg4", "configs[4] shard")):
        d = jload(f"{ROUND}_bench_{tag}.json")
        if d:
            f[tag] = d["value"]
            f[tag + "_scale100"] = d.get("value_scale100")
            if tag == "cfg4":
                f["cfg4_table"] = (d.get("control_table") or {}).get("value")
                ce = d.get("control_exact") or {}
                f["cfg4_exact"] = ce.get("value")
                sc = ce.get("scorer") or {}
                f["cfg4_exact_workers"], f["cfg4_exact_cost_us"], f["cfg4_memo_hit"] = sc.get("workers"), sc.get("cost_us_per_12_word_sentence"), sc.get("memo_hit_frac")
    # pytest -m gpu summary: counts and the worst parity figures per engine precision over the full-size goldens
    path = os.path.join(P, f"{ROUND}_gpu_tests_summary.txt")
    if os.path.exists(path):
        txt = open(path).read()
        m = re.search(r"(\d+) passed(?:, (\d+) skipped)?", txt)
        if m:
            f["tests_passed"], f["tests_skipped"] = int(m.group(1)), int(m.group(2) or 0)
        f["tests_failed"] = int((re.search(r"(\d+) failed", txt) or [0, 0])[1])
        worst = {}
        for name, prec, efin, ecos in re.findall(r"^\s+(full_\w+|refine_vs_split\w*)\s+prec=(\d): ([0-9.e+-]+) ([0-9.e+-]+)", txt, re.M):
            w = worst.setdefault(int(prec), [0.0, 0.0])
            w[0], w[1] = max(w[0], float(efin)), max(w[1], float(ecos))
        for prec, key in ((0, "bf16"), (1, "f32"), (3, "split"), (4, "fp16"), (5, "refine")):
            if prec in worst:
                f[f"err_final_{key}"], f[f"err_cos_{key}"] = worst[prec]
        m = re.search(r"host control scorer under the CLIP tower: ([0-9.]+) ms of host work per ([0-9.]+) ms step\s+-> ([0-9.]+)x", txt)
        if m:
            f["overlap_cost_ms"], f["overlap_step_ms"], f["overlap_ratio"] = float(m.group(1)), float(m.group(2)), float(m.group(3))
    rv = jlines(f"{ROUND}_refine_validate_256x10_fullcaptions.jsonl") or jlines(f"{ROUND}_refine_validate_128x10.jsonl")
    for r in rv:
        if r.get("mode") == "generate" and r.get("gate_delta", 0) > 0:
            f["rv_gen_ids_identical"], f["rv_gen_gated"], f["rv_gen_images"], f["rv_gen_steps"] = r["ids_identical"], r["gated_frac"], r["images"], r["image_steps"]
        elif "max_abs_dfinal" in r:
            f["rv_max_dfinal"], f["rv_p999"], f["rv_winners"], f["rv_image_steps"] = r["max_abs_dfinal"], r["p999"], r["winners_identical"], r["image_steps"]
            f["rv_guard_max_dev"], f["rv_true_max_dev"] = r.get("guard_max_dev"), r.get("true_max_dev")
    # round 6: the further weight draws of the refine validation, and what the de-duplication removes on a trained-like head
    draws = [r for r in jlines(f"{ROUND}_refine_validate_draws_128x10.jsonl")] + [r for r in jlines(f"{ROUND}_refine_validate_draws2_256x10.jsonl")]
    gen = [r for r in draws if r.get("mode") == "generate"]
    stp = [r for r in draws if "max_abs_dfinal" in r and (r.get("draw") or {}).get("outlier_gain", 1.0) != 12.0]
    if gen:
        f["draws_n"] = len(gen)
        f["draws_images"] = sum(r["images"] for r in gen)
        f["draws_identical"] = sum(r["images_with_identical_ids"] for r in gen)
        f["draws_image_steps"] = sum(r["image_steps"] for r in gen)
        f["draws_tripped"] = sum(1 for r in gen if r["guard_tripped_image_steps"] > 0)
    if stp:
        f["draws_worst_dfinal"] = max(r["max_abs_dfinal"] for r in stp)
    div = [d_ for name in (f"{ROUND}_refine_validate_divergences_128x10.jsonl", f"{ROUND}_refine_validate_draws2_256x10.jsonl")
           for r in jlines(name) if r.get("mode") == "generate" for d_ in r.get("divergences", [])]
    if div:
        f["div_n"] = len(div)
        f["div_max_gap"] = max(d_["gap_to_other_engines_choice"] for d_ in div)
    path = os.path.join(P, f"{ROUND}_gpu_tests_summary.txt")
    if os.path.exists(path):
        m = re.findall(r"^\s+prec=(\d) B=(\d+)\s+(\d+)/(\d+) = ([0-9.]+)\s+rows (\d+)/(\d+) = ([0-9.]+)", open(path).read(), re.M)
        for prec, B_, dd, seqs, frac, ra_, rb_, rfrac in m:
            if prec == "0" and B_ == "64":
                f["dedup_seq_frac"], f["dedup_row_frac"] = float(frac), float(rfrac)
    drv8 = jload(f"{ROUND}_bench_8ranks_shared_gpu.json")
    if drv8:
        f["r8_n"], f["r8_backend_ranks"] = drv8.get("n_gpus"), (drv8.get("ranks") or {}).get("reported_by_backend")
        f["r8_broadcast_s"], f["r8_gather_s"] = (drv8.get("ranks") or {}).get("broadcast_s"), (drv8.get("ranks") or {}).get("gather_s")
    return f


def fmt(v, spec=".3g", none="n/a"):
    return none if v is None else format(v, spec)


def block_status(f):
    rows = [
        "| | |",
        "|---|---|",
        f"| `pytest -m gpu` on the MI355X (`profiles/{ROUND}_gpu_tests_summary.txt`) | {fmt(f.get('tests_passed'), 'd')} passed, {fmt(f.get('tests_skipped'), 'd')} skipped, {fmt(f.get('tests_failed'), 'd')} failed |",
        f"| fused score vs the reference, worst over the full-size goldens | bf16 engine {fmt(f.get('err_final_bf16'), '.2e')} (bar 1e-3), screen-then-refine {fmt(f.get('err_final_refine'), '.2e')}, split-fp16 {fmt(f.get('err_final_split'), '.1e')}, f32 {fmt(f.get('err_final_f32'), '.1e')} |",
        f"| headline, BASELINE configs[2] (`profiles/{ROUND}_bench_driver_cmd.json` = the driver's `--gpus 1 --steps 20 --warmup 5`) | **{fmt(f.get('value'), '.1f')} captions/s** bf16 engine (HF-init logit scale), **{fmt(f.get('value_scale100'), '.1f')}** screen-then-refine (published checkpoints' logit scale); one stream {fmt(f.get('single_stream_value'), '.1f')}; this box ran the vendor library's dense bf16 8192^3 matmul at {fmt(f.get('calib'), '.0f')} TFLOP/s (`box_calibration`: the pool's boxes differ by several per cent on identical code, `profiles/r05_tower_ab.txt` holds the same-box A/Bs) |",
        f"| roofline of the CLIP-text GEMM family (dense bf16 peak 2.5 PFLOP/s) | {fmt(f.get('frac'), '.3f')} over the timed two-stream region, {fmt(f.get('frac_1s'), '.3f')} on one stream; `clip_text_mfma_util` (GEMM + attention FLOPs over all CLIP-text kernel time) {fmt(f.get('tower_util'), '.3f')} |",
        f"| screen-then-refine in `czc_generate` | {fmt(None if f.get('gated_frac') is None else 100 * f['gated_frac'], '.0f')} % of the image-steps pass the margin gate, {fmt(None if f.get('re_encoded_frac') is None else 100 * f['re_encoded_frac'], '.1f')} % of the candidates re-encoded; against the all-split engine: ids identical = {f.get('rv_gen_ids_identical', 'n/a')} over {fmt(f.get('rv_gen_steps'), 'd')} image-steps, `czc_step` worst fused-score difference {fmt(f.get('rv_max_dfinal'), '.2e')} over {fmt(f.get('rv_image_steps'), 'd')} image-steps |",
        f"| the same engine on {fmt(f.get('draws_n'), 'd')} further weight draws (`profiles/{ROUND}_refine_validate_draws_128x10.jsonl`, `..._draws2_256x10.jsonl`: eight other seed pairs, x12 / x6 / x3 outlier towers) | `czc_step` worst fused-score difference {fmt(f.get('draws_worst_dfinal'), '.2e')} on the draws that do not trip the guard; free-running 10 sweeps: {fmt(f.get('draws_identical'), 'd')} of {fmt(f.get('draws_images'), 'd')} images ({fmt(f.get('draws_image_steps'), 'd')} image-steps) keep the all-split engine's ids, the {fmt(f.get('div_n'), 'd')} that leave do so where the split engine's own winner and runner-up are {fmt(f.get('div_max_gap'), '.1e')} or less apart (`..._divergences_...`); the x12 tower trips the guard |",
        f"| exact de-duplication on a trained-like MLM head (`test_dedup_is_exact`, 64 images) | {fmt(None if f.get('dedup_seq_frac') is None else 100 * f['dedup_seq_frac'], '.0f')} % of the candidates ride on an identical one, text-tower rows x{fmt(f.get('dedup_row_frac'), '.2f')}; nothing to remove on the flat-softmax synthetic headline |",
        f"| 8 ranks on one shared GPU (`profiles/{ROUND}_bench_8ranks_shared_gpu.json`, gloo rendezvous, test flag) | n_gpus {fmt(f.get('r8_n'), 'd')}, {fmt(f.get('r8_backend_ranks'), 'd')} ranks reported by the backend, weight broadcast {fmt(f.get('r8_broadcast_s'), '.2f')} s, gather {fmt(f.get('r8_gather_s'), '.3f')} s; the multi-GPU curve itself is unmeasured (no node) |",
        f"| other BASELINE shapes (`profiles/{ROUND}_bench_cfg*.json`) | single image {fmt(f.get('cfg1'), '.2f')} captions/s; configs[3] shard (256 images, shuffle, L=15, K=512) {fmt(f.get('cfg3'), '.1f')}; configs[4] shard (64 images, sentiment, L=12): table mode {fmt(f.get('cfg4_table'), '.1f')}, exact host scorer {fmt(f.get('cfg4_exact'), '.1f')} ({fmt(f.get('cfg4_exact_workers'), 'd')} worker interpreters, stand-in tagger at {fmt(f.get('cfg4_exact_cost_us'), '.0f')} us per 12-word sentence) |",
        f"| CPU oracle on the box's host (one full caption, {fmt(f.get('cpu_threads'), 'd')} threads) | {fmt(f.get('cpu_value'), '.4f')} captions/s |",
    ]
    return "\n".join(rows)


# kernel-name pattern -> (what it is in the step, what bounds it, work model)
#   work model: ("gemm", N, Kdim, which rows) FLOPs = 2 * rows * N * Kdim per pass;  ("bytes", bytes per row, which rows);  None
KERNELS = [
    (r"gemm_wreg_kernel<1,", "fc1 512->2048 + quick-GELU, LayerNorm folded in (weights in registers, `gemm_wreg.hip`)", "MFMA", ("gemm", 2048, 512, "mlp")),
    (r"gemm256x_kernel<0, true, (false|true), 256>", "fc2 2048->512 + residual on fp16 rows (256x256 ping-pong ring, `gemm256.hip`)", "MFMA", ("gemm", 512, 2048, "mlp")),
    (r"gemm_wreg_kernel<0,", "q/k/v 512->1536, LayerNorm folded in (weights in registers)", "MFMA", ("gemm", 1536, 512, "all")),
    (r"attention_image_kernel", "branch attention, one work-group per (image, 4 heads) (`attention.hip`)", "HBM: 4 KB per row (q, k, v in, context out)", ("bytes", 4096, "all")),
    (r"gemm_wreg_resid_kernel", "out-projection 512->512 + residual, x updated in place, LayerNorm partials out", "HBM: 3 KB per row (context, x in, x out; the second column group's context read hits L2)", ("gemmbytes", 512, 512, "mlp", 3072)),
    (r"gemm_kernel<czc::split_t, 0, true, false, 128>", "BERT q/k/v and fc2 slices (split-fp16, 3 MFMA passes, 128x128 tiles, `gemm.hip`)", "latency / tile count (180-540 tiles)", None),
    (r"gemm_kernel<czc::split_t, 2, true, false, 128>", "BERT fc1 + GELU", "latency / tile count", None),
    (r"gemm_kernel<czc::split_t, 0, true, true, 64>", "BERT out-projection (64-wide tiles), pruned last layer", "latency", None),
    (r"attention_mfma_split_kernel", "BERT attention (split-fp16)", "latency", None),
    (r"ln_finalize_kernel", "(mean, rstd) per row from the producers' 16 partial sums", "HBM: 136 B per row", ("bytes", 136, "all2")),
    (r"layernorm_kernel<czc::split_t>", "BERT residual + LayerNorm (behind the out-projection)", "HBM", None),
    (r"layernorm_splitk_kernel", "BERT fc2: sum of the K slices + bias + residual -> LayerNorm in one pass", "HBM", None),
    (r"clip_embed_kernel", "token + position embedding -> fp16 rows + LayerNorm statistics of layer 0", "HBM (fp32 table rows in, fp16 rows out)", None),
    (r"bridge_kernel", "WordPiece decode -> CLIP BPE ids, control scores (`bridge.hip`)", "latency", None),
    (r"attention_mfma_kernel", "trunk attention (one wave per image and head)", "latency", None),
    (r"splitk_reduce_kernel", "sum of split-K slices + bias + residual (layers without a LayerNorm behind them)", "HBM", None),
    (r"softmax_mask_topk_kernel", "softmax(logits / tau) * mask -> top-K (`topk.hip`)", "HBM: 122 KB per image", None),
    (r"scan_kernel", "exclusive scan of the segment lengths", "latency (one work-group)", None),
    (r"gemm_kernel<czc::split_t, 0, false, false, 128>", "MLM decoder 768->30522 on the masked rows", "weight streaming", None),
    (r"cosine_kernel", "cosine of every candidate with its image", "HBM", None),
    (r"combine_kernel", "softmax_K, fusion, first argmax, write-back (`combine.hip`)", "latency", None),
    (r"prefix_plan_kernel", "shared-prefix plan + exact de-duplication", "latency", None),
]


def block_kernels(f):
    """what runs in one caption batch of configs[2] on ONE stream, from the rocprofv3 kernel stats of
    `bench.py --streams 1 --steps 2 --warmup 1` (4 caption batches in the process) and the workload's row counts."""
    import csv
    path = os.path.join(P, f"{ROUND}_bench_bf16_kernel_stats.csv")
    dflt = jload(f"{ROUND}_bench_default.json") or jload(f"{ROUND}_bench_driver_cmd.json")
    if not os.path.exists(path) or not dflt:
        return "(no kernel stats collected for this round yet)"
    rows = list(csv.DictReader(open(path)))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    batches = 4
    R = dflt["clip_rows_per_step"]                        # packed text-tower rows of one caption batch (sum over its 100 position-steps)
    S = dflt["config"]["images_per_gpu"] * dflt["config"]["candidate_k"] * dflt["config"]["sentence_len"] * dflt["config"]["num_iterations"]
    which = {"all": 12 * R, "mlp": 11 * R + S, "all2": 23 * R}   # rows a kernel class sees per caption batch (last layer: EOS rows only)
    out = ["| kernel | role | launches per caption batch | mean us | share of GPU time | achieved | bound |", "|---|---|---|---|---|---|---|"]
    shown = 0.0
    for pat, role, bound, work in KERNELS:
        hit = [r for r in rows if re.search(pat, r["Name"])]
        if not hit:
            continue
        ns = sum(float(r["TotalDurationNs"]) for r in hit)
        calls = sum(int(r["Calls"]) for r in hit)
        ach = ""
        if work and work[0] in ("gemm", "gemmbytes"):
            fl = 2.0 * which[work[3]] * work[1] * work[2] * batches
            ach = f"{fl / ns / 1e3:.0f} TFLOP/s = {fl / ns / 1e3 / 2500:.2f} of peak"
            if work[0] == "gemmbytes":
                ach += f"; {which[work[3]] * work[4] * batches / ns / 1e3:.1f} TB/s"
        elif work and work[0] == "bytes":
            ach = f"{which[work[2]] * work[1] * batches / ns / 1e3:.1f} TB/s"
        name = re.sub(r"\(anonymous namespace\)::|czc::|void ", "", hit[0]["Name"]).split("(")[0]
        out.append(f"| `{name}` | {role} | {calls // batches} | {ns / calls / 1e3:.0f} | {100 * ns / total_ns:.1f} % | {ach} | {bound} |")
        shown += ns
    out.append(f"| everything else ({len(rows)} kernel names in all) | | | | {100 * (total_ns - shown) / total_ns:.1f} % | | |")
    out.append("")
    out.append(f"GPU time of one caption batch on one stream: {total_ns / batches / 1e6:.0f} ms over {sum(int(r['Calls']) for r in rows) // batches} launches; "
               f"{R / 1e6:.1f} M packed rows x 12 layers (the reference would run {S * 15 / 1e6:.1f} M).")
    return "\n".join(out)


def block_header(f):
    """The measured figures of the precision comments in include/conzic_hip.h (C comment lines)."""
    lines = [
        f" * Measured on one MI355X, round 6 (generated by tools/refresh_docs.py from profiles/{ROUND}_*):",
        f" *   fused score vs the reference, worst over the full-size goldens: CZC_PREC_BF16 {fmt(f.get('err_final_bf16'), '.2e')}, CZC_PREC_REFINE {fmt(f.get('err_final_refine'), '.2e')},",
        f" *   CZC_PREC_SPLIT {fmt(f.get('err_final_split'), '.1e')}, CZC_PREC_F32 {fmt(f.get('err_final_f32'), '.1e')} (bar 1e-3);",
        f" *   CZC_PREC_REFINE against CZC_PREC_SPLIT over {fmt(f.get('rv_image_steps'), 'd')} more image-steps: worst {fmt(f.get('rv_max_dfinal'), '.2e')}, 99.9th percentile {fmt(f.get('rv_p999'), '.1e')}, winners identical",
        f" *   {fmt(f.get('rv_winners'), 'd')} / {fmt(f.get('rv_image_steps'), 'd')}; guard sample maximum {fmt(f.get('rv_guard_max_dev'), '.2e')} against {fmt(f.get('rv_true_max_dev'), '.2e')} over all candidates;",
        f" *   BASELINE configs[2]: {fmt(f.get('value'), '.1f')} captions/s (CZC_PREC_BF16), {fmt(f.get('value_scale100'), '.1f')} (CZC_PREC_REFINE through czc_generate, {fmt(None if f.get('gated_frac') is None else 100 * f['gated_frac'], '.0f')} % of the image-steps gated).",
    ]
    return "\n".join(lines)


TARGETS = [
    ("README.md", "status", block_status, "<!-- {} GENERATED {} -->"),
    ("DESIGN.md", "status", block_status, "<!-- {} GENERATED {} -->"),
    ("DESIGN.md", "kernels", block_kernels, "<!-- {} GENERATED {} -->"),
    ("include/conzic_hip.h", "measured", block_header, " * {} GENERATED {}"),
]


def apply(check):
    f = facts()
    stale = []
    for rel, name, fn, marker in TARGETS:
        path = os.path.join(ROOT, rel)
        txt = open(path).read()
        b, e = marker.format("BEGIN", name), marker.format("END", name)
        if b not in txt or e not in txt:
            stale.append(f"{rel}: markers of block '{name}' missing")
            continue
        i, j = txt.index(b) + len(b), txt.index(e)
        new = txt[:i] + "\n" + fn(f) + "\n" + txt[j:]
        if new != txt:
            if check:
                stale.append(f"{rel}: block '{name}' is stale (run tools/refresh_docs.py)")
            else:
                open(path, "w").write(new)
    return stale


if __name__ == "__main__":
    problems = apply("--check" in sys.argv)
    for p_ in problems:
        print(p_)
    sys.exit(1 if problems else 0)
