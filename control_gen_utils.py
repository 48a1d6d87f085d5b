"""Drop-in for the reference's control_gen_utils.py (sentiment and POS control; signatures as in
control_gen_utils.py:30-33, 82-85, 136-139, 197-200).

The reference scores every candidate sentence through nltk on the host (sentiments_classifer.py:9-48,
POS_classifier.py:6-31) and fuses `gamma * softmax_K(score)` (+ the repeat penalty) into the step's score
(control_gen_utils.py:53-59, :160-168).  Here the fusion runs inside the score-combine kernel and the raw scores come from
(conzic_amd/control.py, chosen per call, `CZC_CONTROL=auto|exact|table`):

* the reference's own sentence arithmetic on the decoded candidate strings, called back from the engine once per step
  (`czc_set_control_callback`) -- the default wherever nltk imports: captions equal the reference's;
* per-BERT-token tables evaluated inside the text-bridge kernel (`CZC_CONTROL=table`: built once per tokenizer from nltk; or
  handed over by the caller as `clip.lexicon` / `clip.lexicon_pos` / `clip.pos_tags`): no host work per step, a context-free
  approximation of the tagger."""
import time

from conzic_amd.runtime import run_generation


def sentiment_sequential_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                                    max_len=15, top_k=0, temperature=None, alpha=0.7, beta=1,
                                    max_iters=20, batch_size=1, verbose=True, gamma=5, ctl_signal="positive"):
    """control_gen_utils.py:30-80"""
    return run_generation("sequential", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose, gamma=gamma,
                          ctl_signal=ctl_signal)


def sentiment_shuffle_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                                 max_len=15, top_k=0, temperature=None, alpha=0.7, beta=1,
                                 max_iters=20, batch_size=1, verbose=True, gamma=5, ctl_signal="positive"):
    """control_gen_utils.py:82-134"""
    return run_generation("shuffle", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose, gamma=gamma,
                          ctl_signal=ctl_signal)


def POS_sequential_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                              max_len=15, top_k=0, temperature=None, alpha=0.7, beta=1, gamma=0.1,
                              max_iters=20, batch_size=1, ctl_signal=["DET"], verbose=True):
    """control_gen_utils.py:136-195"""
    logger.info(ctl_signal)
    return run_generation("sequential", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose, gamma=gamma,
                          pos_template=ctl_signal)


def control_generate_caption(img_name, model, clip, tokenizer, image_instance, token_mask, logger,
                             prompt="", batch_size=10, max_len=25,
                             top_k=100, temperature=1.0, max_iter=500, alpha=0.7, beta=1, gamma=5,
                             ctl_type="sentiment", style_type="positive", pos_type=None, generate_order="sequential"):
    """control_gen_utils.py:197-232"""
    start_time = time.time()
    if ctl_type == "sentiment":
        fn = sentiment_sequential_generation if generate_order == "sequential" else sentiment_shuffle_generation
        generate_texts, clip_scores = fn(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                                         batch_size=batch_size, max_len=max_len, top_k=top_k, alpha=alpha, beta=beta,
                                         gamma=gamma, temperature=temperature, max_iters=max_iter, ctl_signal=style_type)
    else:  # POS control (control_gen_utils.py:218-223)
        generate_texts, clip_scores = POS_sequential_generation(
            img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger, batch_size=batch_size,
            max_len=max_len, top_k=top_k, alpha=alpha, beta=beta, gamma=gamma, temperature=temperature,
            ctl_signal=pos_type, max_iters=max_iter)
    logger.info("Finished in %.3fs" % (time.time() - start_time))
    final_caption = generate_texts[-2]
    best_caption = generate_texts[-1]
    for i in range(batch_size):
        logger.info(f"The {i + 1}-th image: {img_name[i]}")
        logger.info(f"final caption: {final_caption[i]}")
        logger.info(f"best caption: {best_caption[i]}")
    return generate_texts, clip_scores
