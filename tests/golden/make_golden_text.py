#!/usr/bin/env python3
"""Golden vectors for the frozen context text encoder, from the REAL reference class `CLIPTextContextEncoder`
(/root/reference/models/_modified_BiomedCLIP.py:798-883 over `Transformer` / `ResidualAttentionBlock`, :371-431).  Same import
recipe and stand-ins as make_golden_attn.py (none of them is on this class's forward path: DropPath is nn.Identity at drop_path 0).
Small configuration (2 layers, width 64) plus one at the reference's real shape (context 42, width 512, 8 heads, 12 layers);
weights rebuilt from seeds (attn_fixture_util.seeded_state), so the fixture holds token ids, contexts and outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_text.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_fixture_util import TEXT_CASES, seeded_state  # noqa: E402
from make_golden_attn import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_golden.npz")


def main():
    ref = load_reference()
    out = {}
    g = torch.Generator().manual_seed(424242)
    with torch.no_grad():
        for tag, (kw, n_ctx, K, B, seed) in TEXT_CASES.items():
            m = ref.CLIPTextContextEncoder(**kw).eval()
            m.load_state_dict(seeded_state(m, seed))
            N1 = kw["context_length"] - n_ctx
            text = torch.randint(1, kw["vocab_size"] - 1, (K, N1), generator=g)
            for k in range(K):  # end-of-text = the largest id, at a different position per prompt
                eot = 2 + (k * 3) % (N1 - 2)
                text[k, eot] = kw["vocab_size"] - 1
                text[k, eot + 1:] = 0
            context = torch.randn(B, n_ctx, kw["transformer_width"], generator=g) * 0.5
            out[f"{tag}/text"], out[f"{tag}/context"] = text.numpy(), context.numpy()
            out[f"{tag}/out"] = m(text, context).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
