"""Drop-in for the reference's gen_utils.py: same function names, positional/keyword arguments
and return structure (gen_utils.py:51-53, 98-101, 148-150, 197-198, 289-292), executed by the
MI355X-native engine (libconzic_hip.so) instead of eager torch.

    generate_texts, clip_scores = generate_caption(img_name, model, clip, tokenizer, image_instance,
                                                   token_mask, logger, prompt=..., batch_size=..., ...)

`parallel_generation` is deliberately absent: it is unreachable from the reference's CLI
(`--order` choices, demo.py:33) and indexes with the wrong variable (gen_utils.py:265).
"""
import time

from conzic_amd.runtime import run_generation


def sequential_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len=15, top_k=100, temperature=None, alpha=0.7, beta=1,
                          max_iters=20, batch_size=1, verbose=True):
    """Generate one word at a time, in L->R order (gen_utils.py:51-96)."""
    return run_generation("sequential", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose)


def shuffle_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                       max_len=15, top_k=0, temperature=None, alpha=0.7, beta=1,
                       max_iters=20, batch_size=1, verbose=True):
    """Generate one word at a time, in random generation order (gen_utils.py:98-146)."""
    return run_generation("shuffle", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose)


def span_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                    max_len=15, top_k=0, temperature=None, alpha=0.7, beta=1,
                    max_iters=20, batch_size=1, verbose=True):
    """Generate span_len=2 words per BERT forward, in L->R order (gen_utils.py:148-195)."""
    return run_generation("span", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose)


def random_generation(img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                      max_len=15, top_k=0, temperature=None, alpha=0.7, beta=2, max_iters=300, print_every=10,
                      batch_size=1, verbose=True):
    """Generate for one random position per step (gen_utils.py:197-242)."""
    return run_generation("random", img_name, model, clip, tokenizer, image_instance, token_mask, prompt, logger,
                          max_len, top_k, temperature, alpha, beta, max_iters, batch_size, verbose,
                          print_every=print_every)


def generate_caption(img_name, model, clip, tokenizer, image_instance, token_mask, logger,
                     prompt="", batch_size=1, max_len=15,
                     top_k=100, temperature=1.0, max_iter=500, alpha=0.7, beta=1,
                     generate_order="sequential"):
    """Main entry point (gen_utils.py:289-333)."""
    start_time = time.time()
    common = dict(batch_size=batch_size, max_len=max_len, top_k=top_k, alpha=alpha, beta=beta, temperature=temperature)
    if generate_order == "sequential":
        generate_texts, clip_scores = sequential_generation(img_name, model, clip, tokenizer, image_instance,
                                                            token_mask, prompt, logger, max_iters=max_iter, **common)
    elif generate_order == "shuffle":
        generate_texts, clip_scores = shuffle_generation(img_name, model, clip, tokenizer, image_instance, token_mask,
                                                         prompt, logger, max_iters=max_iter, **common)
    elif generate_order == "random":
        max_iter *= max_len                                     # gen_utils.py:305-306
        generate_texts, clip_scores = random_generation(img_name, model, clip, tokenizer, image_instance, token_mask,
                                                        prompt, logger, max_iters=max_iter, print_every=max_len,
                                                        verbose=True, **common)
    elif generate_order == "span":
        generate_texts, clip_scores = span_generation(img_name, model, clip, tokenizer, image_instance, token_mask,
                                                      prompt, logger, max_iters=max_iter, **common)
    else:
        raise ValueError(f"generate_order must be sequential|shuffle|random|span, got {generate_order!r}")
    logger.info("Finished in %.3fs" % (time.time() - start_time))
    final_caption = generate_texts[-2]
    best_caption = generate_texts[-1]
    for i in range(batch_size):
        logger.info(f"The {i + 1}-th image: {img_name[i]}")
        logger.info(f"final caption: {final_caption[i]}")
        logger.info(f"best caption: {best_caption[i]}")
    return generate_texts, clip_scores
