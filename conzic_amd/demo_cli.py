"""Harness in the role of the reference's demo.py / run.py (the callers of the boundary):
same option names and defaults (demo.py:15-76), same sequence of calls (demo.py:105-153):
set_seed once -> load LM / tokenizer / CLIP -> build token_mask from stop words -> loop samples
calling generate_caption / control_generate_caption.

    python -m conzic_amd.demo_cli --synthetic --run_type caption --order sequential
    python -m conzic_amd.demo_cli --lm_model <dir> --match_model <dir> --caption_img_path img.jpg ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--device", type=str, default='cuda', choices=['cuda'])
    p.add_argument('--run_type', default='controllable', nargs='?', choices=['caption', 'controllable'])
    p.add_argument('--prompt', default='Image of a', type=str)
    p.add_argument('--order', default='shuffle', nargs='?', choices=['sequential', 'shuffle', 'span', 'random'])
    p.add_argument('--control_type', default='sentiment', nargs='?', choices=["sentiment", "pos"])
    # demo.py:40-45 declares this with type=list (unusable from a shell); here: a JSON list of tag lists
    p.add_argument('--pos_type', type=json.loads,
                   default=[['DET'], ['ADJ', 'NOUN'], ['NOUN'], ['VERB'], ['VERB'], ['ADV'], ['ADP'], ['DET', 'NOUN'],
                            ['NOUN'], ['NOUN', '.'], ['.', 'NOUN'], ['.', 'NOUN']],
                   help="predefined part-of-speech template (JSON)")
    p.add_argument('--sentiment_type', default="positive", nargs='?', choices=["positive", "negative"])
    p.add_argument('--samples_num', default=2, type=int)
    p.add_argument("--sentence_len", type=int, default=10)
    p.add_argument("--candidate_k", type=int, default=200)
    p.add_argument("--alpha", type=float, default=0.02)
    p.add_argument("--beta", type=float, default=2.0)
    p.add_argument("--gamma", type=float, default=5.0)
    p.add_argument("--lm_temperature", type=float, default=0.1)
    p.add_argument("--num_iterations", type=int, default=10)
    p.add_argument("--lm_model", type=str, default='bert-base-uncased')
    p.add_argument("--match_model", type=str, default='openai/clip-vit-base-patch32')
    p.add_argument("--caption_img_path", type=str, default=None)
    p.add_argument("--stop_words_path", type=str, default=None)
    p.add_argument("--synthetic", action="store_true", help="random-init weights + synthetic vocab/images (no checkpoints)")
    p.add_argument("--tiny", action="store_true", help="with --synthetic: tiny model dims")
    p.add_argument("--control_scores", default=None, choices=["auto", "table", "exact"],
                   help="controllable runs: the reference's own nltk sentence scorer called back once per step (exact; what "
                        "auto picks when nltk imports) or per-token tables built from nltk and evaluated inside the engine's "
                        "kernels (table: the throughput mode, context-free approximation); sets CZC_CONTROL")
    a = p.parse_args(argv)
    if a.control_scores:
        os.environ["CZC_CONTROL"] = a.control_scores
    return a


def main(argv=None):
    args = get_args(argv)
    import utils
    from clip.clip import CLIP
    from control_gen_utils import control_generate_caption
    from gen_utils import generate_caption
    from conzic_amd import synth
    from conzic_amd.models import SyntheticLM
    from conzic_amd.text import tokenizers_from_vocab
    import logging

    utils.set_seed(args.seed)
    logger = logging.getLogger("ConZIC")
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    if args.synthetic:
        sv = synth.make_vocab_tiny() if args.tiny else synth.make_vocab()
        bcfg = synth.bert_tiny(len(sv.bert_tokens)) if args.tiny else synth.bert_base()
        ccfg = synth.clip_tiny(len(sv.clip_vocab)) if args.tiny else synth.clip_b32()
        lm_tokenizer, clip_tok = tokenizers_from_vocab(sv)
        lm_model = SyntheticLM(bcfg)
        clip = CLIP.from_state(ccfg, synth.make_clip_weights(ccfg, 12), clip_tok)
        from conzic_amd import control
        if control.import_nltk() is None:
            # no nltk: synthetic per-token control tables (with nltk the runtime builds the tables from it, as it does
            # for real checkpoints -- conzic_amd/control.py; --control_scores exact calls the reference's scorer per step)
            clip.lexicon = synth.make_lexicon(len(sv.bert_tokens))
            clip.pos_tags = synth.make_pos_tags(len(sv.bert_tokens))
        token_mask = synth.make_token_mask(sv)
        from PIL import Image
        images = [Image.fromarray(u) for u in synth.make_images_u8(args.batch_size, ccfg.v_image)]
    else:
        from transformers import AutoModelForMaskedLM, AutoTokenizer
        lm_model = AutoModelForMaskedLM.from_pretrained(args.lm_model).eval()
        lm_tokenizer = AutoTokenizer.from_pretrained(args.lm_model)
        clip = CLIP(args.match_model)
        with open(args.stop_words_path, 'r', encoding='utf-8') as f:          # demo.py:135-143
            stop_words = [w.rstrip('\n') for w in f.readlines()]
        token_mask = np.ones((1, lm_tokenizer.vocab_size), dtype=np.float32)
        for sid in lm_tokenizer.convert_tokens_to_ids(stop_words):
            token_mask[0, sid] = 0
        from PIL import Image
        images = [Image.open(args.caption_img_path).convert("RGB")]
    image_instance = images if args.batch_size > 1 else images[0]
    img_name = [f"img{j}" for j in range(args.batch_size)]
    t0 = time.time()
    for sample_id in range(args.samples_num):                                   # demo.py:83 (no reseeding)
        logger.info(f"Sample {sample_id}: ")
        kw = dict(prompt=args.prompt, batch_size=args.batch_size, max_len=args.sentence_len, top_k=args.candidate_k,
                  temperature=args.lm_temperature, max_iter=args.num_iterations, alpha=args.alpha, beta=args.beta,
                  generate_order=args.order)
        if args.run_type == 'caption':
            generate_caption(img_name, lm_model, clip, lm_tokenizer, image_instance, token_mask, logger, **kw)
        else:
            control_generate_caption(img_name, lm_model, clip, lm_tokenizer, image_instance, token_mask, logger,
                                     gamma=args.gamma, ctl_type=args.control_type, style_type=args.sentiment_type,
                                     pos_type=args.pos_type, **kw)
    logger.info("total %.2fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
