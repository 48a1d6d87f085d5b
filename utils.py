"""Drop-in for the reference's utils.py (same names, arguments and behaviour):
create_logger :8-35, set_seed :37-44, get_init_text :46-51, update_token_mask :53-59,
format_output :61-74.  Host-side plumbing only; nothing here is on the GPU hot path."""
import logging
import os
import random

import numpy as np


def create_logger(folder, filename):
    """Logger 'ConZIC': stream handler + file handler with '%(message)s' (utils.py:8-35).
    colorlog is optional here (the reference hard-requires it only to colour an empty format)."""
    logger = logging.getLogger('ConZIC')
    logging.root.setLevel(logging.DEBUG)
    stream = logging.StreamHandler()
    stream.setLevel(logging.DEBUG)
    stream.setFormatter(logging.Formatter(""))
    os.makedirs(folder, exist_ok=True)
    hdlr = logging.FileHandler(os.path.join(folder, filename))
    hdlr.setLevel(logging.DEBUG)
    hdlr.setFormatter(logging.Formatter("%(message)s"))
    logger.addHandler(hdlr)
    logger.addHandler(stream)
    return logger


def set_seed(seed):
    """random / numpy / torch seeds (utils.py:37-44); the shuffle and random visiting orders are
    drawn from these process-global streams exactly as in the reference."""
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
            torch.cuda.manual_seed_all(seed)
    except ImportError:
        pass


def get_init_text(tokenizer, seed_text, max_len, batch_size=1):
    """Initial sentence: seed_text padded with [MASK] to max_len (utils.py:46-51)."""
    text = seed_text + tokenizer.mask_token * max_len
    ids = tokenizer.encode(text)
    return [ids] * batch_size


def update_token_mask(tokenizer, token_mask, max_len, index):
    """'.' is only allowed in the last position (utils.py:53-59); mutates token_mask in place."""
    token_mask[:, tokenizer.vocab['.']] = 1 if index == max_len - 1 else 0
    return token_mask


def format_output(sample_num, FinalCaption, BestCaption):
    """UI string formatting (utils.py:61-74): first min(sample_num, 5) captions, newline-joined."""
    n = sample_num if 1 <= sample_num <= 4 else 5
    return "\n".join(f"{c}" for c in FinalCaption[:n]), "\n".join(f"{c}" for c in BestCaption[:n])
