"""A/B of the 64-wide small-grid form of the tiled GEMM (test option gemm_small_tiles = N: taken when the 128-wide tiles of a
launch number at most N/4 of the CUs) at small batches.  usage: small_tiles_ab.py IMAGES N [N ...]"""
import json, os, runpy, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from conzic_amd import native
    native.load_test().czc_test_set_option(b"gemm_small_tiles", int(sys.argv[3]))
    sys.argv = ["bench.py", "--images", sys.argv[2], "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-invariance", "--no-profile"]
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
else:
    images = sys.argv[1]
    for rnd in range(2):
        for n in sys.argv[2:]:
            out = subprocess.run([sys.executable, __file__, "--one", images, n], capture_output=True, text=True).stdout.strip().splitlines()
            d = json.loads(out[-1])
            print(f"images={images} gemm_small_tiles={n}: {d['value']:.3f} captions/s", flush=True)
