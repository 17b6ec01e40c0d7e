#!/usr/bin/env python3
"""Golden vectors for the BiomedCLIP context text encoder, from the REAL reference class `HFContextTextEncoder`
(/root/reference/models/_modified_BiomedCLIP.py:885-1015) -- dev container only; writes data only: tests/golden/hftext_golden.npz.

What is real and what cannot run.  The class is instantiated from the reference file (import recipe of make_golden_attn.py, except
that the two HF helper modules it imports, models/BiomedCLIP/BiomedCLIP/hf_configs.py and hf_model.py, are loaded from the
reference too instead of being stood in: the pooler and `arch_dict` ARE on this class's path).  Its constructor (hard-coded
PubMedBERT config :909-918, `ClsLastHiddenStatePooler`, MLP projection), `token_embedding` (:950-958), the attention-mask
construction (:966-969), pooler and projection (:979-980) are the reference's own code.  The one step that cannot run here is
`modified_BertModel.forward` (:1081-1191): a copy of transformers-4.x `BertModel.forward` that calls `self.get_head_mask`, which
the installed transformers 5.15 no longer has (AttributeError).  The reference pins no transformers version (SURVEY.md section 8c
lists it as third-party arithmetic outside /root/reference), so that step is taken from the dependency itself: the parent class's
`BertModel.forward(inputs_embeds=, attention_mask=)` of the installed transformers, on the reference object's own weights -- the
function the reference's override restates (its extra `context` argument is unused, :1159-1165).  No stand-in is written for it.

Weights are not stored: generator and tests rebuild them from seeds (attn_fixture_util.seeded_state).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hftext.py
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_attn as G  # noqa: E402
from attn_fixture_util import HFTEXT_CASES, HFTEXT_SCALE, seeded_state  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hftext_golden.npz")
HF_DIR = "/root/reference/models/BiomedCLIP/BiomedCLIP"


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_with_hf():
    G.load_reference()  # installs the timm stand-ins and the private package skeleton (and a first import with HF stand-ins)
    load_by_path(G.PKG + ".BiomedCLIP.BiomedCLIP.hf_configs", os.path.join(HF_DIR, "hf_configs.py"))
    load_by_path(G.PKG + ".BiomedCLIP.BiomedCLIP.hf_model", os.path.join(HF_DIR, "hf_model.py"))
    return load_by_path(G.PKG + "._modified_BiomedCLIP", G.REF)  # re-executed: now binds the real ClsPooler / _POOLERS / arch_dict


def main():
    from transformers import BertModel
    ref = load_reference_with_hf()
    out = {}
    g = torch.Generator().manual_seed(515151)
    with torch.no_grad():
        for tag, (K, N1, N2, seed) in HFTEXT_CASES.items():
            m = ref.HFContextTextEncoder().eval()
            m.load_state_dict(seeded_state(m, seed, scale=HFTEXT_SCALE))
            x = torch.randint(1000, 30000, (K, N1), generator=g)
            x[:, 0] = 2  # [CLS]
            for k in range(K):  # [SEP] then padding, a different length per prompt
                end = 3 + (k * 2) % (N1 - 3)
                x[k, end] = 3
                x[k, end + 1:] = 0
            context = torch.randn(1, N2, 768, generator=g) * 0.05
            # reference forward (:960-991), its transformer call replaced by the parent BertModel.forward (see the header)
            attn_mask = torch.ones((5, N2 + x.shape[1])).long()
            mask_t = (x != m.config.pad_token_id).long()
            attn_mask[:, 0:1] = mask_t[:, 0:1]
            attn_mask[:, N2 + 1:] = mask_t[:, 1:]
            hs = BertModel.forward(m.transformer, inputs_embeds=m.token_embedding(x, context), attention_mask=attn_mask)
            projected = m.proj(m.pooler(hs, attn_mask))
            out[f"{tag}/x"], out[f"{tag}/context"] = x.numpy(), context.numpy()
            out[f"{tag}/out"] = projected.numpy()
            out[f"{tag}/cls_hidden"] = hs.last_hidden_state[:, 0].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
