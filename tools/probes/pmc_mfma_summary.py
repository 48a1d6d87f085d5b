"""Summarise the rocprofv3 --pmc pass of tools/probes/r03_profiles.sh (SQ_INSTS_VALU_MFMA_MOPS_{BF16,F16}, MFMA busy cycles,
GRBM_GUI_ACTIVE over ONE caption batch of bench.py on one stream): counter-derived MFMA FLOPs per kernel family (one MOPS
count = 512 FLOPs: a 32x32x16 MFMA = 32768 FLOPs = 64 counts) against the FLOPs the engine itself counts
(bench.py executed_tflop_per_caption x images), and MFMA-busy share of the SIMD cycles per family.
usage: pmc_mfma_summary.py <dir with *counter_collection.csv>"""
import csv
import glob
import json
import sys
from collections import defaultdict

FAMILIES = [("gemm_rowln", "gemm_rowln_kernel"), ("gemm256x", "gemm256x_kernel"), ("gemm256q", "gemm256q_kernel"),
            ("gemm256sq", "gemm256sq_kernel"), ("gemm_wreg_resid", "gemm_wreg_resid_kernel"), ("gemm_wreg", "gemm_wreg_kernel"), ("gemm_128_split", "gemm_kernel<czc::split_t"),
            ("gemm_128_bf16", "gemm_kernel<unsigned short"), ("gemm_skinny", "gemm_skinny"), ("attention", "attention_")]


def family(name):
    for fam, key in FAMILIES:
        if key in name:
            return fam
    return "other"


def main(d):
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                fam = family(row["Kernel_Name"])
                per[fam][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    launches[fam] += 1
    out = {}
    tot = 0.0
    for fam, c in per.items():
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)
        fl = 512.0 * mops
        tot += fl
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[fam] = dict(launches=launches[fam], mfma_tflop=round(fl / 1e12, 3),
                        mfma_busy_share_of_simd_cycles=None if not gui else round(busy / (1024.0 * gui / 8.0), 4),  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                        gui_active_cycles=gui)
    clip = sum(v["mfma_tflop"] for k, v in out.items() if k in ("gemm_rowln", "gemm256x", "gemm256q", "gemm_wreg", "gemm_wreg_resid", "gemm_128_bf16"))
    print(json.dumps(dict(per_family=out, total_mfma_tflop=round(tot / 1e12, 3), clip_half_precision_gemm_tflop=round(clip, 3),
                          files=len(files)), indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
