"""Does running two half-batches concurrently (two engines = two streams, one GPU) hide one half's BERT / small kernels
behind the other half's CLIP tower?    Times configs[2] as ONE engine on 256 images against
TWO engines on 128 images each driven from two host threads (ctypes releases the GIL), same total work.
usage: two_stream_probe.py [images=256] [passes=2]"""
import sys
import threading
import time

sys.path.insert(0, __file__.rsplit('/', 3)[0])
import torch  # noqa: E402
from conzic_amd import harness, native, synth  # noqa: E402
from conzic_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, K, I, seed_len = 10, 200, 10, 4
dev = torch.device("cuda", 0)
pos, nm, every = harness.order_positions("sequential", L, I)
hp = Engine.hyper(0.02, 2.0, 0.1)


def make(n_images, first):
    bcfg, ccfg = synth.bert_base(), synth.clip_b32()
    su = harness.build_synthetic(False, native.PREC_BF16, logit_scale=2.6592, regular_only=True, device=0,
                                 bert_w=synth.make_bert_weights(bcfg, 11), clip_w=synth.make_clip_weights(ccfg, 12),
                                 bert_cfg=bcfg, clip_cfg=ccfg)
    pix = torch.from_numpy(synth.pixels_from_u8(synth.make_images_u8(n_images, first=first))).to(dev)
    init = su.bert_tok.encode("Image of a" + su.bert_tok.mask_token * L)

    def step():
        su.engine.encode_images(pix)
        return su.engine.generate(n_images, init, L, seed_len, K, pos, hp, n_mask=nm, snapshot_every=every)
    return su, step


def timed(steps):
    for s in steps:
        s()  # warm-up, one after the other
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=lambda s=s: [s() for _ in range(PASSES)]) for s in steps]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


su1, one = make(B, 0)
dt1 = timed([one])
print(f"one engine, {B} images: {B * PASSES / dt1:.2f} captions/s", flush=True)
su1.engine.close()
for n in (2, 3, 4):
    if B % n and n != 3:
        continue
    sizes = [B // n + (1 if r < B % n else 0) for r in range(n)]
    engs, first = [], 0
    for sz in sizes:
        engs.append(make(sz, first))
        first += sz
    dtn = timed([e[1] for e in engs])
    print(f"{n} engines on {n} streams, {sizes} images: {B * PASSES / dtn:.2f} captions/s", flush=True)
    if n == 2:
        dts = sum(timed([e[1]]) for e in engs)
        print(f"the same two engines one after the other: {B * PASSES / dts:.2f} captions/s", flush=True)
    for e in engs:
        e[0].engine.close()
