"""Harness in the role of the reference's run.py (directory of images, batches, JSON results):
same batching semantics -- `os.listdir` order, `batch_size`, `drop_last=True` (run.py:156-178) --
and the same on-disk layout (run.py:194-222):

    results/<run>_<order>_len<L>_topk<K>_alpha<a>_beta<b>_gamma<g>_lmTemp<t>/sample_<n>/iter_<k>.json
    .../best_clipscore.json        each a {image_name: caption} dict

    python -m conzic_amd.run_cli --synthetic --caption_img_path <dir> --run_type caption --order shuffle
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from conzic_amd.demo_cli import get_args  # same options as demo.py/run.py (run.py:15-76)


def result_dir(args, run_type, sample_id):
    """run.py:196-197 / :210-211"""
    prefix = "caption" if args.run_type == "caption" else run_type
    return "results/%s_%s_len%d_topk%d_alpha%.3f_beta%.3f_gamma%.3f_lmTemp%.3f/sample_%d" % (
        prefix, args.order, args.sentence_len, args.candidate_k, args.alpha, args.beta, args.gamma,
        args.lm_temperature, sample_id)


def merge_results(all_results, gen_texts, names):
    """run.py:86-92: all_results[iter_id][image_name] = caption (last entry = best-by-CLIP caption)."""
    for iter_id, texts in enumerate(gen_texts):
        if all_results[iter_id] is None:
            all_results[iter_id] = {}
        for name, text in zip(names, texts):
            all_results[iter_id][name] = text
    return all_results


def write_results(save_dir, all_results):
    """run.py:198-207"""
    os.makedirs(save_dir, exist_ok=True)
    for iter_id, res in enumerate(all_results):
        fn = "best_clipscore.json" if iter_id == len(all_results) - 1 else f"iter_{iter_id}.json"
        with open(os.path.join(save_dir, fn), "w") as f:
            json.dump(res, f)


def batches(names, batch_size):
    """DataLoader(shuffle=False, drop_last=True) over os.listdir order (run.py:158-178)."""
    for s in range(0, len(names) - batch_size + 1, batch_size):
        yield names[s:s + batch_size]


def main(argv=None):
    args = get_args(argv)
    import logging
    import numpy as np
    from PIL import Image
    import utils
    from clip.clip import CLIP
    from control_gen_utils import control_generate_caption
    from gen_utils import generate_caption
    from conzic_amd import synth
    from conzic_amd.models import SyntheticLM
    from conzic_amd.text import tokenizers_from_vocab

    utils.set_seed(args.seed)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger("ConZIC")
    run_type = "caption" if args.run_type == "caption" else args.sentiment_type
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
    else:
        from transformers import AutoModelForMaskedLM, AutoTokenizer
        lm_model = AutoModelForMaskedLM.from_pretrained(args.lm_model).eval()
        lm_tokenizer = AutoTokenizer.from_pretrained(args.lm_model)
        clip = CLIP(args.match_model)
        with open(args.stop_words_path, 'r', encoding='utf-8') as f:
            stop_words = [w.rstrip('\n') for w in f.readlines()]
        token_mask = np.ones((1, lm_tokenizer.vocab_size), dtype=np.float32)
        for sid in lm_tokenizer.convert_tokens_to_ids(stop_words):
            token_mask[0, sid] = 0
    img_dir = args.caption_img_path
    names = os.listdir(img_dir)
    # One process per GPU (torchrun): batches are block-partitioned over the ranks (conzic_amd/dist.py); every rank
    # walks ALL (sample, batch) pairs in the reference's order and advances the order RNG for the batches it skips,
    # so an N-rank run produces the single-process run's captions batch for batch.  No collective while polishing;
    # one gather of the caption dicts at the end of each sample.
    from conzic_amd import dist as czd
    from conzic_amd.runtime import advance_order_rng
    from clip.clip import ImageEmbeds
    rank, world, local = czd.env_rank_world()
    if world > 1:
        import torch
        import torch.distributed as tdist
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            # host resources of this rank: the CPUs of its GPU's NUMA node, torch threads and scorer workers to match
            logger.info(f"rank {rank}: host share {czd.pin_rank(local, device_index=local % torch.cuda.device_count())}")
        if not tdist.is_initialized():
            tdist.init_process_group(os.environ.get("CZC_DIST_BACKEND", "nccl"))
    all_batches = list(batches(names, args.batch_size))
    own_lo, own_hi = czd.shard_range(len(all_batches), rank, world)
    embed_cache = {}  # batch index -> image_embeds [B, proj]: the ViT runs once per image, not once per sample
    for sample_id in range(args.samples_num):
        all_results = [None] * (args.num_iterations + 1)
        logger.info(f"Sample {sample_id + 1}: ")
        for batch_idx, name_batch in enumerate(all_batches):
            if not (own_lo <= batch_idx < own_hi):
                advance_order_rng(args.order, args.sentence_len, args.num_iterations)
                continue
            logger.info(f"The {batch_idx + 1}-th batch:")
            if batch_idx in embed_cache:
                imgs = ImageEmbeds(embed_cache[batch_idx])
            else:
                imgs = [Image.open(os.path.join(img_dir, n)).convert("RGB") for n in name_batch]
            kw = dict(prompt=args.prompt, batch_size=args.batch_size, max_len=args.sentence_len,
                      top_k=args.candidate_k, temperature=args.lm_temperature, max_iter=args.num_iterations,
                      alpha=args.alpha, beta=args.beta, generate_order=args.order)
            if args.run_type == 'caption':
                gen_texts, _ = generate_caption(name_batch, lm_model, clip, lm_tokenizer, imgs, token_mask, logger, **kw)
            else:
                gen_texts, _ = control_generate_caption(name_batch, lm_model, clip, lm_tokenizer, imgs, token_mask,
                                                        logger, gamma=args.gamma, ctl_type=args.control_type,
                                                        style_type=args.sentiment_type, pos_type=args.pos_type, **kw)
            if batch_idx not in embed_cache:
                embed_cache[batch_idx] = clip.last_image_embeds()
            all_results = merge_results(all_results, gen_texts, name_batch)
        if world > 1:
            import torch.distributed as tdist
            parts = [None] * world
            tdist.all_gather_object(parts, all_results)
            all_results = [None] * (args.num_iterations + 1)
            for part in parts:  # rank order == batch order
                for it, d in enumerate(part):
                    if d is not None:
                        all_results[it] = {**(all_results[it] or {}), **d}
        if rank == 0:
            write_results(result_dir(args, run_type, sample_id), all_results)


if __name__ == "__main__":
    main()
