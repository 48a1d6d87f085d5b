import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """Measured parity figures of the GPU tests (tests/test_step_gpu.py::ERR_LOG / DIVERGENCE_LOG / REFINE_LOG), so that a
    GPU run leaves the numbers DESIGN.md quotes in its log."""
    cmod = sys.modules.get("test_control_gpu")
    flips = getattr(cmod, "FLIPS", None) if cmod else None
    if flips:
        terminalreporter.write_line("controllable path against the reference's own scorers over a context-dependent tagger "
                                    "(case, mode: winners that differ / image-steps, worst |d final_score|):")
        for name, mode, fl, steps, worst in flips:
            terminalreporter.write_line(f"  {name:22s} {mode:6s}: {fl}/{steps}   {worst:.3e}")
    for step_ms, cost_ms, ratio in (getattr(cmod, "OVERLAP", None) or []) if cmod else []:
        terminalreporter.write_line(f"host control scorer under the CLIP tower: {cost_ms:.2f} ms of host work per {step_ms:.2f} ms step "
                                    f"-> {ratio:.3f}x the wall time of a free scorer")
    mod = sys.modules.get("test_step_gpu")
    if not mod:
        return
    log = getattr(mod, "ERR_LOG", None)
    if log:
        worst = {}
        for name, prec, efin, ecos in log:
            w = worst.setdefault((name, prec), [0.0, 0.0, 0])
            w[0], w[1], w[2] = max(w[0], efin), max(w[1], ecos), w[2] + 1
        terminalreporter.write_line("golden parity, worst over image-steps (case, precision: |d final_score|, |d cosine|, n):")
        for (name, prec), (efin, ecos, n) in sorted(worst.items()):
            terminalreporter.write_line(f"  {name:22s} prec={prec}: {efin:.3e} {ecos:.3e} n={n}")
    dlog = getattr(mod, "DIVERGENCE_LOG", None)
    if dlog:
        terminalreporter.write_line("free-running half-precision engines vs the reference trajectory (case, precision: identical tokens "
                                    "while on the trajectory, images still on it at the end, reference top-2 margin at each first divergence):")
        for name, prec, tot, same, alive, B, margins in dlog:
            terminalreporter.write_line(f"  {name:22s} prec={prec}: {same}/{tot} tokens, {alive}/{B} images, margins {[f'{m:.1e}' for m in margins]}")
    glog = getattr(mod, "GUARD_LOG", None)
    if glog:
        terminalreporter.write_line("screen-then-refine guard (case: max |screening error - mean| on re-encoded candidates, tripped image-steps; budget 2.5e-4):")
        for name, dev, trips in glog:
            terminalreporter.write_line(f"  {name:22s} {dev:.3e}  tripped {trips}")
    olog = getattr(mod, "OUTLIER_LOG", None)
    if olog:
        terminalreporter.write_line("screen-then-refine engine on towers with outlier LayerNorm channels (gain factor: worst |d final_score| vs the split "
                                    "engine, guard max_dev, tripped / image-steps):")
        for f_, worst, dev, trips, n in olog:
            terminalreporter.write_line(f"  x{f_:<5g} {worst:.3e}  {dev:.3e}  {trips}/{n}")
    slog = getattr(mod, "SOFT_LOG", None)
    if slog:
        terminalreporter.write_line("tiny towers, half-precision engines: steps whose winner flipped at a near-tie (case precision: flips / steps): "
                                    + ", ".join(f"{n_} {p_} {s_}/{t_}" for n_, p_, s_, t_ in slog))
    ddlog = getattr(mod, "DEDUP_LOG", None)
    if ddlog:
        terminalreporter.write_line("exact de-duplication on a trained-like MLM head (precision, images: candidates riding on an identical one / all, "
                                    "text-tower rows with / without the option):")
        for prec, B_, dd, seqs, ra_, rb_ in ddlog:
            terminalreporter.write_line(f"  prec={prec} B={B_:<3d} {dd}/{seqs} = {dd / max(seqs, 1):.3f}   rows {ra_}/{rb_} = {ra_ / max(rb_, 1):.3f}")
    wlog = getattr(mod, "DRAW_LOG", None)
    if wlog:
        terminalreporter.write_line("screen-then-refine heuristics on other weight draws, against the all-split engine (draw: czc_step worst "
                                    "|d final_score| over untripped image-steps, czc_generate guard max_dev, images with identical ids, guard trips):")
        for name, worst, dev, same, n, trips in wlog:
            terminalreporter.write_line(f"  {name:34s} {worst:.3e}  {dev:.3e}  {same}/{n}  tripped {trips}")
    rlog = getattr(mod, "REFINE_LOG", None)
    if rlog:
        terminalreporter.write_line("screen-then-refine engine: candidate sequences re-encoded by the split-fp16 tower (case: seqs, rows):")
        for name, seqs, rseqs, rows, rrows in rlog:
            terminalreporter.write_line(f"  {name:22s} {rseqs}/{seqs} = {rseqs / max(seqs, 1):.3f}   rows {rrows}/{rows} = {rrows / max(rows, 1):.3f}")
